cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_mlp3_gpu.py tests/test_chain_gpu.py tests/test_render_gpu.py tests/test_render_forward_gpu.py -x -q 2>&1 | tail -3
