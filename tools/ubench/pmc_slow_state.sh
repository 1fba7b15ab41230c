#!/bin/bash
# Measurement aid: counter passes over tools/ubench/placement_timeline (two hipMalloc states + one on 1 GiB granules, timed in
# turn) -- one counter set per process, as the guide asks.  The first pass doubles as the test whether this box draws a slow
# first allocation (>= 12 ms): if it does not, the remaining passes are skipped (exit 3) and the caller tries another box.
R=$PWD
out=$R/gpurun_out/r5/pmc_slow
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  n=$1; shift
  rm -rf $out/$n
  timeout -k 10 120 rocprofv3 --pmc "$@" --kernel-trace -d $out/$n -o tl --output-format json -- $R/tools/ubench/placement_timeline 6 2 1024 > $out/$n.log 2>&1
  echo "pass $n rc=$? $(grep round $out/$n.log | tail -1)"
}
pass wr TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL
if ! grep round $out/wr.log | tail -1 | grep -q '"malloc[01]": 1[2-9]'; then echo "no slow state on this box"; exit 3; fi
pass tag TCC_TAG_STALL TCC_BUBBLE TCC_LATENCY_FIFO_FULL TCC_SRC_FIFO_FULL
pass hit TCC_HIT TCC_MISS TCC_READ TCC_WRITE
pass req TCC_REQ TCC_STREAMING_REQ TCC_NC_REQ TCC_RW_REQ
pass ev TCC_NORMAL_EVICT TCC_NORMAL_WRITEBACK TCC_WRITEBACK TCC_PROBE
pass tcp TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
pass tcp2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LEVEL_WAVES
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE GRBM_EA_BUSY
