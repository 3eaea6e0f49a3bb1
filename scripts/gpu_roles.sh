# the role launch (gact_roles.hpp) against the one-wave-does-all launch: parity tests, in-kernel stamps of both, interleaved bench runs
# usage: TAG=r05_roles [WORKLOADS="ecoli10x pacbio50mb"] [REPS=2] [TESTS="tests/test_gpu_chain.py ..."] bash scripts/gpu_roles.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-r05_roles}
mkdir -p $OUT
if [ -n "${TESTS-tests/test_gpu_chain.py tests/test_gpu_golden.py tests/test_gpu_properties.py}" ]; then
  timeout -k 10 900 python -m pytest ${TESTS-tests/test_gpu_chain.py tests/test_gpu_golden.py tests/test_gpu_properties.py} -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
fi
if [ -f $R/build/libgact_hip_stamps.so ] && [ -z "$NO_STAMPS" ]; then
  for w in ${STAMP_WORKLOADS:-ecoli10x}; do
    env GACT_HIP_${MODE_ENV:-ROLES}=1 GACT_STAMPS_LIB=$R/build/libgact_hip_stamps.so timeout -k 10 200 python $R/tools/stamps.py $w > $OUT/stamps_${w}_roles.txt 2>&1; echo "== stamps $w roles"; cat $OUT/stamps_${w}_roles.txt | tail -22
    GACT_HIP_ROLES=0 GACT_STAMPS_LIB=$R/build/libgact_hip_stamps.so timeout -k 10 200 python $R/tools/stamps.py $w > $OUT/stamps_${w}_old.txt 2>&1; echo "== stamps $w old"; cat $OUT/stamps_${w}_old.txt | tail -22
  done
fi
CASES="${CASES:-roles|darwin-gpu_amd/libgact_hip.so|GACT_HIP_ROLES=1;old|darwin-gpu_amd/libgact_hip.so|X=1}" WORKLOADS="${WORKLOADS:-ecoli10x pacbio50mb}" REPS=${REPS:-2} TAG=${TAG:-r05_roles}/ab bash $R/scripts/gpu_env_ab.sh
