# the reference CLI's full -test self-test (tool/zultra.c:465-641) linked against the round's final build, on the GPU box; the log goes to profiles/r06_full_selftest.txt
O=gpurun_out/r06f; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
{
echo "# ZULTRA_SLOW_TESTS=1 python -m pytest tests/test_reference_cli.py -m gpu -k full_selftest on the MI355X box, build $(python -c 'import zultra_amd; print(zultra_amd.csrc_digest())')"
( time ZULTRA_SLOW_TESTS=1 timeout 3300 python -m pytest tests/test_reference_cli.py -m gpu -k full_selftest -q ) 2>&1 | tail -8
} > $O/full_selftest.txt 2>&1
cat $O/full_selftest.txt
