echo "== lscpu"; lscpu | grep -i "numa\|socket\|model name\|^CPU(s)"
echo "== status"; grep -i "allowed_list" /proc/self/status
echo "== nodes"; ls /sys/devices/system/node/ | tr '\n' ' '; echo
for n in /sys/devices/system/node/node*; do echo $n $(cat $n/cpulist) $(grep MemTotal $n/meminfo) $(grep MemFree $n/meminfo); done
echo "== gpus"; for d in /sys/class/drm/card*/device; do echo $d $(cat $d/numa_node 2>/dev/null) $(cat $d/vendor 2>/dev/null) $(readlink -f $d | sed 's/.*\///'); done
echo "== kfd"; for t in /sys/class/kfd/kfd/topology/nodes/*; do echo $t $(grep -E "simd_count|drm_render_minor|location_id" $t/properties | tr '\n' ' '); done 2>/dev/null | head -20
echo "== rocm-smi topo"; rocm-smi --showtoponuma 2>/dev/null | head -30
echo "== visible"; echo HIP_VISIBLE_DEVICES=$HIP_VISIBLE_DEVICES ROCR_VISIBLE_DEVICES=$ROCR_VISIBLE_DEVICES
nproc; free -g | head -2
cat /sys/kernel/mm/transparent_hugepage/enabled
