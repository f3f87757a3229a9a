"""Import-only placeholder for OpenCV (absent from this image), on its own path so that nothing else ever sees it: it lets
vggt/modules/utils_warp.py (which does `import cv2` at the top) be imported by tools/make_goldens.py for the code paths that never call
OpenCV (forward warping with fill_cracks=False).  Any use of an OpenCV function raises."""


def __getattr__(name):
    raise RuntimeError(f"cv2.{name} was called: OpenCV is not available here; this placeholder only satisfies `import cv2`")
