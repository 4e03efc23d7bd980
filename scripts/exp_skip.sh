cd openimucameracalibrator_amd/csrc
for mask in 0x0000 0x1110 0x8880 0x7770 0xfff0; do
  rm -f kernels_cholesky.o
  make -s HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -DOICC_DBG_SKIP=$mask" liboicc_hip.so 2>&1 | grep -i error
  echo "mask $mask"; (cd ../..; python scripts/prof_solver.py 5 2>&1 | tail -2)
done
