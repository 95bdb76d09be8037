timeout 300 python tools/dev_latmb.py 2,3,8,12,13,16,17,24,30,32 2>&1 | grep -v amdgpu.ids | tail -14
