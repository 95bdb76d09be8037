run() { echo "## $*"; timeout 600 "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d.get('ok'), d.get('seconds'), [(m['call'],m['cycle'],m['rows_differing'],m['pred_differing'],m['ranges'][:6]) for m in d.get('mismatches',[])][:4])"; }
for i in 1 2 3 4; do
run python tools/guard_stress.py --precision fp32 --cycles 3 --guard 1 --sizes 600,1281,3072,4100 --max-mismatches 99
run python tools/guard_stress.py --precision fp32 --cycles 3 --guard 1 --sizes 600,1281,3072,4100 --max-mismatches 99 --device-io
done
export DCE_GUARD_VA_REUSE=1
echo "address ranges given back and reused"
for i in 1 2 3 4; do
run python tools/guard_stress.py --precision fp32 --cycles 3 --guard 1 --sizes 600,1281,3072,4100 --max-mismatches 99
done
