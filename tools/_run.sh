set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -s -k "iou_labels or supervised_step_matches or two_ranks_share or graph_replay" > $O/pytest_new.log 2>&1; echo "rc=$?" >> $O/pytest_new.log
