cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_entrypoint.py tests/test_gpu_session.py -x -q -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -30 $O/pytest.log
