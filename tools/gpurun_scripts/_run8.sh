export WF_LIB=$GRAFT_REPO_ROOT/worldforge_amd/_lib/libwf_hip_convablate.so
for d in 0 1 2 3 7; do echo "WF_CONV_DEBUG=$d"; WF_CONV_DEBUG=$d python - <<'PY'
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import conv_bench as cb
for x3 in (False, True):
    cb.bench(81,480,832,96,96,1,x3)
    cb.bench(41,120,208,384,384,1,x3)
PY
done 2>&1 | grep -v amdgpu
