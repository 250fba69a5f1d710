python tools/exp/dbg_pair.py 2>&1 | grep -v amdgpu.ids
echo "---- kMaxSeg=4 build"
touch od_wscl_amd/csrc/gemm_bf16.hip
ODW_EXTRA_FLAGS=-DODW_MAX_SEG=4 python -c "from od_wscl_amd import _build; _build.build()" 
python tools/exp/dbg_pair.py 2>&1 | grep -v amdgpu.ids
