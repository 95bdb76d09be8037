# round 6: what do the LDS bank-conflict replays of conv_h2's write-backs cost?  The product kernel against a timing probe whose
# un-pooled write-backs store from eight lanes of sixteen (same instructions, no two-way conflict, WRONG results): time per launch by HIP events, board power and
# clock, and the conflict counters of both.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
X=$R/deep_contact_estimator_amd/libdce_experiments.so
P=$R/deep_contact_estimator_amd/libdce_h2exp.so
for L in $X $P $X $P; do
  DCE_LIB=$L python $R/bench.py --precision fp32_f16x2 --steps 400 --warmup 50 --no-cpu-baseline --no-extras > /tmp/ab.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/ab.json')); print('$(basename $L)', round(d['value']/1e6,3), 'M windows/s', {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})"
done
for L in $X $P; do
  DCE_LIB=$L rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_$(basename $L) -o p --output-format csv -- python $R/bench.py --precision fp32_f16x2 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("/tmp/pmc_$(basename $L)/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:28]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    if "conv_h2" in k:
        m = {c: x / cnt[(k, c)] for c, x in v.items()}
        print("$(basename $L)", k, {c: round(x) for c, x in m.items()}, "conflict / LDS-active = %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]))
PY
done
(DCE_LIB=$X python $R/bench.py --precision fp32_f16x2 --steps 4000 --warmup 50 --no-cpu-baseline --no-extras --no-kernel-timing > /dev/null 2>&1 &) ; sleep 6; rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | head -4; sleep 4
(DCE_LIB=$P python $R/bench.py --precision fp32_f16x2 --steps 4000 --warmup 50 --no-cpu-baseline --no-extras --no-kernel-timing > /dev/null 2>&1 &) ; sleep 6; rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | head -4; sleep 4
