cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_eval_helper.py -q -m gpu -x -k "nms or eval" 2>&1 | tail -n 3
