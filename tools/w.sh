cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p tools/bin
g++ -O2 -std=c++17 tools/hostpath_bench.cpp -o tools/bin/hostpath_bench -pthread 2>/dev/null
taskset -c 64-71 tools/bin/hostpath_bench | grep -v "rep [0-3]" | grep -A3 "decide each batch ahead\|4 threads write" | grep -v "^--" | cut -c1-250
python tools/dense_probe.py 2>&1 | grep -v "amdgpu" | grep -v "^gate\|^noise"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -m gpu -x -q 2>&1 | tail -3
