#!/bin/bash
# Collects PMC counters for the bench's kernels, one counter group per pass
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE cannot share a pass; no trace domains
# besides --kernel-trace next to --pmc).
# Groups 6 and 7 are the L2's memory-side request counters by size (gfx950 has a 128-byte class that the shipped
# FETCH_SIZE expression tallies as 64 B): read bytes = 32 n32 + 64 n64 + 128 n128 exactly, no calibration factor.
# usage: PMC_GROUPS="1 2 3" tools/pmc_passes.sh <outdir> [bench args...]
set -u
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
groups=(
  "FETCH_SIZE"
  "WRITE_SIZE"
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"
  "SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE"
  "TCC_HIT_sum TCC_MISS_sum"
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
  "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum"
  "GRBM_TA_BUSY"  # the TA_* block counters hang rocprofv3 on this pool (two 300 s timeouts): not collected
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
  "TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
)
for i in ${PMC_GROUPS:-1 2 3 4 5}; do
  grp=${groups[$((i-1))]}
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/pass$i" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-probe --no-end-to-end --sustain-seconds 0 --schedule serial "$@" > "$out/pass$i.log" 2>&1
  echo "pass $i ($grp): exit $?"
done
SVC_PMC_BENCH_ARGS="--schedule serial $*" python3 tools/summarize_pmc.py "$out" "$out/summary.csv"
rm -rf "$out"/pass*/  # raw per-dispatch CSVs are large; the summary is what is kept
