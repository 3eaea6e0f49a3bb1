# one counter set over several builds of the library: LIBS="default build/libgact_hip_x.so ..." PASSES="A B" CMD=... TAG=...
set -e
R=$GRAFT_REPO_ROOT
for lib in $LIBS; do
  name=$(basename $lib .so)
  if [ "$lib" = default ]; then envv="X=1"; else envv="GACT_HIP_LIB_PATH=$R/$lib"; fi
  echo "== $name"
  ENVV="$envv" TAG=$TAG/$name bash $R/scripts/gpu_pmc_any.sh
done
