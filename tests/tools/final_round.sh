mkdir -p gpurun_out/r05v
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05v/pytest.log 2>&1; echo rc=$? >> gpurun_out/r05v/pytest.log; tail -4 gpurun_out/r05v/pytest.log
TAG=r05z OTHER_SHAPES=1 bash tests/tools/profile_round.sh > gpurun_out/r05z_profile.log 2>&1; tail -6 gpurun_out/r05z_profile.log | cut -c1-200
