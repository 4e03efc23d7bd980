cd openimucameracalibrator_amd/csrc
for mode in 0 1 2; do
  rm -f kernels_blocks.o
  make -s HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -DOICC_DBG_ATOMICS=$mode" liboicc_hip.so 2>&1 | grep -i " error"
  echo "mode $mode"; (cd ../..; python - <<'PY'
import sys
sys.path.insert(0, '.')
from openimucameracalibrator_amd import synthetic, estimator as E
F = E.SPLINE | E.T_I_C | E.GRAVITY_DIR
for cfg in ("C2", "C5"):
    ds = synthetic.make_config(cfg)
    cal = E.ImuCameraCalibrator().BatchInitSpline(ds)
    p, k = cal.trajectory_.TimeJacobianPass(F, repeats=5)
    print(cfg, "pass ms %.4f" % p, "view/accel/gyro", [round(x, 4) for x in k])
PY
)
done
rm -f kernels_blocks.o; make -s liboicc_hip.so
