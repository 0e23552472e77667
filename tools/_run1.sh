cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r02/gpu_tests17.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r02/gpu_tests17.log
