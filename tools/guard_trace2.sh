run() { echo "## $*"; timeout 600 "$@" 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-1500; }
run python tools/guard_stress.py --precision fp32 --cycles 4 --guard 1 --sizes 1281,3072,4100
run python tools/guard_stress.py --precision fp32 --cycles 4 --guard 1 --sizes 1281,3072,4100 --device-io
run python tools/guard_stress.py --precision fp32 --cycles 4 --guard 1 --sizes 30,200,600 --sequence 0
