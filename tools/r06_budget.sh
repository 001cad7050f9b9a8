O=gpurun_out/r06; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python tools/step_budget.py > $O/step_budget.txt 2>&1; tail -2 $O/step_budget.txt
