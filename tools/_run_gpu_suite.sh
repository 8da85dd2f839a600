set -x
timeout 3000 python -m pytest tests -x -q -m gpu --timeout 900 2>&1 | tail -15
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python tools/kbench.py --sizes n18,nd --iters 30 --extra 2>&1 | grep -i "proximal\|mask_topk"
