#!/bin/bash
# Probe the GPU box for what the test.mp4 / cv2 parity legs need.  Output -> gpurun_out/probe.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/probe.txt
mkdir -p "$R/gpurun_out"
{
  echo "== python modules"
  for m in cv2 imageio imageio_ffmpeg av decord skvideo albumentations PIL torchvision torchcodec moviepy; do
    python - <<PY 2>&1 | tail -1
try:
    import $m
    print("$m: OK", getattr($m, "__version__", "?"))
except Exception as e:
    print("$m: MISSING (%s)" % type(e).__name__)
PY
  done
  echo "== binaries"
  for b in ffmpeg ffprobe gst-launch-1.0 mplayer vlc; do printf "%s: " $b; which $b || echo "not found"; done
  echo "== libs"
  ldconfig -p | grep -i -E "avcodec|avformat|openh264|x264|gstreamer|libva|vcn|rocdecode" || echo "no av libs in ldconfig"
  ls /opt/rocm/lib | grep -i -E "rocdecode|rocjpeg" || echo "no rocdecode in /opt/rocm/lib"
  echo "== cpu"
  lscpu | grep -E "Model name|^CPU\(s\)|Socket|Thread|Core"
  python -c "import os; print('os.cpu_count', os.cpu_count())"
  echo "== gpu"
  rocm-smi --showproductname 2>/dev/null | head -12
  rocminfo | grep -E "gfx|Compute Unit" | head -6
} > "$O" 2>&1
cat "$O"
