cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g12
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/g12/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/g12/pytest.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
