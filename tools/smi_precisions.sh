for p in fp32 fp32_f16x2 fp32_split bf16_fc; do bash tools/smi_watch.sh gpurun_out/r5z_smi_$p.log python bench.py --precision $p --steps 6000 --warmup 50 --no-cpu-baseline --no-extras --no-kernel-timing > /dev/null 2>&1; python - $p <<'PY'
import sys, statistics
rows = []
for l in open("gpurun_out/r5z_smi_%s.log" % sys.argv[1]):
    a = l.split()
    try: rows.append((float(a[0].strip("()Mhz")), float(a[1])))
    except Exception: pass
busy = [r for r in rows if r[1] > 600]
if busy: print(sys.argv[1], "samples", len(busy), "sclk median %.0f MHz (min %.0f max %.0f)" % (statistics.median(r[0] for r in busy), min(r[0] for r in busy), max(r[0] for r in busy)), "power median %.0f W (max %.0f)" % (statistics.median(r[1] for r in busy), max(r[1] for r in busy)))
else: print(sys.argv[1], rows[:5])
PY
done
