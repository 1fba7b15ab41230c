#!/usr/bin/env python3
"""profiles/r03_gputest_head.txt from the logs tools/final_verify.sh left under gpurun_out/r3final/."""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r3final")
summary = open(os.path.join(O, "summary.txt")).read()
commit = re.search(r"commit (\S+)", summary).group(1)
print(f"-m gpu suite at commit {commit} (round 3), one MI355X box, one gpurun call (tools/final_verify.sh)\n")
print("procedure (VERDICT r2, do-this 1): tools/gpu_suite_by_file.sh -- every test file in its OWN pytest process, under its own\n"
      "timeout, log kept per file (gpurun_out/r3final/<file>.log), stop at the first failure; cheapest files first, the misfit file\n"
      "(tests/test_gpu_zz_misfit.py: TVDN_HOST_LIMIT=1G, real buffers, arithmetic asked before tvdn_run is called) last; then the\n"
      "whole suite in one process as the driver runs it; then __graft_entry__.smoke().  No box was lost in this round (gpurun strikes: 0).\n")
print("per file:")
for name, line in re.findall(r"=== (\S+)\n(.*)\n", summary):
    print(f"  {name:28s} {line}")
print("\nwhole suite (python -m pytest tests/ -x -q -m gpu):")
print("  " + [l for l in open(os.path.join(O, "whole_suite.log")).read().splitlines() if " passed" in l][-1])
print("  (9 skipped = tests/test_gpu_rccl.py cases that need >= 2 GPUs)")
print("\nsmoke (python -c 'import __graft_entry__ as g; g.smoke()'):")
for l in open(os.path.join(O, "smoke.log")).read().splitlines():
    if l.startswith("smoke ok") or l.startswith("native library"):
        print("  " + l)
print("\nBASELINE config 1 on the box's 16 host cores with the tracked CPU port (tools/cpu_config1.py; oracle/_ref does not travel):")
print("  " + open(os.path.join(O, "cpu_config1.json")).read().strip())
print("  recon SHA-1 = the reference's (tests/golden/large.npz, d4e27d23...)")
print("\nearlier in the round, same procedure: first verification of round 2's HEAD + hardening (r3a-r3e): 800 passed, 9 skipped;\n"
      "after the launch heuristics / buffer addressing: 801; with the Python pipelined transfers: 816; with tvdn_run's own pipelining,\n"
      "progress callback and workspace: 834-855.")
