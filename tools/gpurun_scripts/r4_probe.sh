#!/bin/bash
# Ask the GPU box which third-party packages of the reference's hot path it has (cv2, pytorch3d, diffusers, flash_attn ...).
#   gpurun --timeout 300 -- 'bash tools/gpurun_scripts/r4_probe.sh'   -> gpurun_out/r4_probe/probe.txt (+ g20/g21/g22 if recordable)
mkdir -p gpurun_out/r4_probe
O=gpurun_out/r4_probe/probe.txt
: > $O
for m in cv2 pytorch3d diffusers flash_attn xformers torchvision imageio skimage kornia; do
  python3 - "$m" >> $O 2>&1 <<'PY'
import importlib, sys
m = sys.argv[1]
try:
    mod = importlib.import_module(m)
    print(m, "PRESENT", getattr(mod, "__version__", "?"), getattr(mod, "__file__", "?"))
except Exception as e:
    print(m, "ABSENT", type(e).__name__, str(e)[:120])
PY
done
echo "--- pip list (grep)" >> $O
python3 -m pip list 2>/dev/null | grep -i -E "opencv|pytorch3d|diffusers|flash|xformers|torchvision|kornia|scikit-image|imageio" >> $O
echo "--- find cv2 / libopencv on disk" >> $O
find / -xdev \( -name "cv2*" -o -name "libopencv*" -o -name "opencv*" -o -name "pytorch3d*" \) -not -path "/proc/*" 2>/dev/null | head -20 >> $O
echo "--- wheelhouse" >> $O
find / -xdev -name "*.whl" -not -path "/proc/*" 2>/dev/null | grep -i -E "opencv|pytorch3d|diffusers" | head >> $O
echo "--- rocm-smi / host" >> $O
nproc >> $O; rocm-smi --showproductname 2>/dev/null | head -8 >> $O
if grep -q "^cv2 PRESENT" $O; then
  python3 tools/record_thirdparty_goldens.py > gpurun_out/r4_probe/record.log 2>&1
  cp tests/golden/g20_farneback_cv2.npz tests/golden/g21_crackfill_cv2.npz gpurun_out/r4_probe/ 2>/dev/null
fi
if grep -q "^pytorch3d PRESENT" $O; then
  python3 tools/record_thirdparty_goldens.py pointrender >> gpurun_out/r4_probe/record.log 2>&1
  cp tests/golden/g22_pointrender_pytorch3d.npz gpurun_out/r4_probe/ 2>/dev/null
fi
cat $O
