# the round's closing run on one box: full GPU suite, then the profiling round on the final sources (copy gpurun_out/$TAG/* to profiles/)
T=${TAG:-r05zz}
mkdir -p gpurun_out/${T}_tests
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_tests/pytest.log 2>&1; echo rc=$? >> gpurun_out/${T}_tests/pytest.log; tail -4 gpurun_out/${T}_tests/pytest.log
TAG=$T OTHER_SHAPES=1 bash tests/tools/profile_round.sh > gpurun_out/${T}_profile.log 2>&1; tail -8 gpurun_out/${T}_profile.log | cut -c1-200
