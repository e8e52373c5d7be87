cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
i=0
for set in "MemUnitBusy MemUnitStalled WriteUnitStalled" "L2CacheHit TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc -o p$i -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/pmc/log$i.txt 2>&1 || echo "set $i failed: $set"
done
ls gpurun_out/pmc
