#!/bin/bash
# Counters of luma_pyr1_kernel<true,128,32> next to the plain 3 : 1 streaming probe (stream_probe_kernel<3,1>: the same read / write mix,
# one unit per short-lived workgroup) in ONE process, one counter group per pass: what the luma kernel does more of per byte than a kernel
# that streams at the mix's rate.  usage (GPU box): tools/pmc_luma_vs_probe.sh <outdir>
set -u
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
groups=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"
  "SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE"
  "TCC_HIT_sum TCC_MISS_sum"
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
  "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
  "TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
)
i=0
for grp in "${groups[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/pass$i" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --sustain-seconds 0 --schedule serial > "$out/pass$i.log" 2>&1
  echo "pass $i ($grp): exit $?"
done
python3 tools/summarize_pmc.py "$out" "$out/summary.csv"
rm -rf "$out"/pass*/
grep -i "luma_pyr1_kernel<true\|stream_probe_kernel<3, 1>" "$out/summary.csv"
