O=gpurun_out/r04d; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 600 python bench.py --gpus 1 --scaling strong --steps 10 --warmup 3 --no-other-configs --no-synthetic > $O/bench_strong.json 2> $O/bench_strong.err; echo "rc=$?" >> $O/bench_strong.err
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-other-configs --no-synthetic > $O/bench_weak.json 2> $O/bench_weak.err; echo "rc=$?" >> $O/bench_weak.err
tail -5 $O/pytest.log; tail -3 $O/bench_strong.err; cut -c1-300 $O/bench_strong.json; python - <<PY
import json
for f in ("bench_strong","bench_weak"):
    try:
        d=json.load(open("$O/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["scaling"], d.get("per_rank"), d["config"]["parallelism"])
    except Exception as e: print(f, "ERR", e)
PY
