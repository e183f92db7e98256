#!/bin/bash
# round-6 evidence: full GPU test suite, bench lines (e2e with cpu_baseline + parity + backbone precision, hot path,
# u8-h2d, --history 3, hi-res, training), steady-state kernel trace, PMC counters of the hand-written hot-path kernels
# (one counter set per pass, --kernel-trace only), SCA traffic json keyed on the kernel source digest.
# Summaries -> gpurun_out/ (copied into profiles/ by hand).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r06_final}
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/${T}_tests.log 2>&1; tail -3 gpurun_out/${T}_tests.log
( time timeout 900 python bench.py --steps 30 --warmup 5 ) > gpurun_out/${T}_bench_e2e.log 2>&1; grep '^{' gpurun_out/${T}_bench_e2e.log | cut -c1-200
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_hot.log 2>&1; grep '^{' gpurun_out/${T}_bench_hot.log | cut -c1-160
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --input u8-h2d > gpurun_out/${T}_bench_e2e_u8.log 2>&1; grep '^{' gpurun_out/${T}_bench_e2e_u8.log | cut -c1-160
timeout 600 python bench.py --steps 10 --warmup 3 --history 3 --no-cpu-baseline > gpurun_out/${T}_bench_e2e_hist3.log 2>&1; grep '^{' gpurun_out/${T}_bench_e2e_hist3.log | cut -c1-160
timeout 600 python bench.py --config configs/occ_hires_400x400x32.py --scope hotpath --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_bench_hires_hot.log 2>&1; grep '^{' gpurun_out/${T}_bench_hires_hot.log | cut -c1-160
timeout 600 python bench.py --mode train --steps 6 --warmup 3 --passes 3 --no-cpu-baseline > gpurun_out/${T}_bench_train.log 2>&1; grep '^{' gpurun_out/${T}_bench_train.log | cut -c1-220
OCC_MSDA_BWD_DETERMINISTIC=1 timeout 600 python bench.py --mode train --steps 6 --warmup 3 --passes 3 --no-cpu-baseline > gpurun_out/${T}_bench_train_det.log 2>&1; grep '^{' gpurun_out/${T}_bench_train_det.log | cut -c1-220
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_e2e -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-extras > $GRAFT_REPO_ROOT/gpurun_out/${T}_trace.log 2>&1)
DB=$(find /tmp/prof_e2e -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 60 --last-ms 60 > gpurun_out/${T}_e2e_kernel_trace_stats.txt 2>&1; head -24 gpurun_out/${T}_e2e_kernel_trace_stats.txt | cut -c1-150
KR="linear_chain|linear_wgrad|sca_fused|tsa_fused|conv3d_mfma|conv3d_bf16x3|conv3d_heads|occ_heads|linear_bf16x3|linear_mfma|value_proj|value_range|point_sampling|conv1x1_nhwc|conv3x3_nhwc|bottleneck64|stem_conv7x7"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TA_TA_BUSY_sum TA_BUSY_avr" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv --kernel-include-regex "$KR" -d /tmp/pmc_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --passes 1 --no-cpu-baseline --no-kernel-timing --no-extras > $GRAFT_REPO_ROOT/gpurun_out/${T}_pmc_$i.log 2>&1)
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${T}_pmc_${i}_counters.csv
done
python tools_dev/make_traffic_json.py "sca_fused" gpurun_out/${T}_sca_gather_traffic.json "profiles/${T}_pmc_derived.txt (rocprofv3 --pmc, one counter set per pass with --kernel-trace only, bench.py e2e scope, base config; gfx950 correction per MI355X_MICROARCH.md: HBM read bytes = 2 x FETCH_SIZE x 1024)" gpurun_out/${T}_pmc_*_counters.csv > /dev/null
python - $T > gpurun_out/${T}_pmc_derived.txt <<'PY'
import csv, glob, collections, sys
T = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sorted(glob.glob(f'gpurun_out/{T}_pmc_[0-9]*_counters.csv')):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        a = acc[k][row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
print(f"# derived per-kernel means (rocprofv3 --pmc, one counter set per pass, --kernel-trace only; bench.py e2e scope, {T})")
print("# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES/32 * 1024); TA = TA_BUSY_avr / (SQ_BUSY_CYCLES/32);")
print("# L2hit = TCC_HIT/(TCC_HIT+TCC_MISS); L2missMB = TCC_MISS*128/1e6; L1req/acc = TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES;")
print("# LDScf = SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE; wait / stall / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES;")
print("# FETCH/WRITE_SIZE in KB as reported (gfx950: HBM read bytes = 2 x FETCH_SIZE x 1024)")
print(f"{'kernel':44s} {'n':>4s} {'kcyc':>8s} {'Mfma':>6s} {'TA':>6s} {'L2hit':>6s} {'L2missMB':>9s} {'L1rq/ac':>8s} {'LDScf':>6s} {'wait':>5s} {'stall':>5s} {'activ':>5s} {'FETCH':>10s} {'WRITE':>10s}")
div = lambda a, b: a / b if b else float('nan')
for k, c in sorted(acc.items()):
    m = lambda n: c[n][1] / c[n][0] if n in c and c[n][0] else float('nan')
    dur = m('SQ_BUSY_CYCLES') / 32
    hit, miss = m('TCC_HIT_sum'), m('TCC_MISS_sum')
    wc = m('SQ_WAVE_CYCLES')
    print(f"{k[:44]:44s} {c['SQ_BUSY_CYCLES'][0]:4d} {dur/1e3:8.1f} {div(m('SQ_VALU_MFMA_BUSY_CYCLES'), dur*1024):6.3f} "
          f"{div(m('TA_BUSY_avr'), dur):6.3f} {div(hit, hit+miss):6.3f} {miss*128/1e6:9.1f} {div(m('TCP_TCC_READ_REQ_sum'), m('TCP_TOTAL_CACHE_ACCESSES_sum')):8.3f} "
          f"{div(m('SQ_LDS_BANK_CONFLICT'), m('SQ_LDS_IDX_ACTIVE')):6.3f} {div(m('SQ_WAIT_ANY'), wc):5.2f} {div(m('SQ_WAIT_INST_ANY'), wc):5.2f} {div(m('SQ_ACTIVE_INST_ANY'), wc):5.2f} "
          f"{m('FETCH_SIZE'):10.0f} {m('WRITE_SIZE'):10.0f}")
PY
cat gpurun_out/${T}_pmc_derived.txt | cut -c1-190
cat gpurun_out/${T}_sca_gather_traffic.json | head -30
