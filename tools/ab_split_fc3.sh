for t in "" "x3_fc3=1"; do DCE_TUNE=$t python bench.py --precision fp32_split --steps 300 --warmup 50 --no-cpu-baseline --no-extras > gpurun_out/ab.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/ab.json'));print('fp32_split $t', round(d['value']/1e6,3), {k:(round(v['avg_ms']*1e3,1), v['launches']) for k,v in d['kernels'].items()})"; done
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in "x3_fc3=0" "x3_fc3=1"; do DCE_TUNE=$t rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_x3_$t -o ab -- python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras --no-kernel-timing --precision fp32_split > /dev/null 2>&1; python - "$t" <<'PY'
import csv, glob, sys
for f in glob.glob("gpurun_out/ab_x3_%s/*kernel_stats.csv" % sys.argv[1]):
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) > 100: print(sys.argv[1], "%-64s calls %5s avg %8.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
