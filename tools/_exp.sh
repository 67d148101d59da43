for v in 0 16 24 32 16 32; do echo "== virt $v"; timeout 100 python tools/train_prof.py 128 10 block 1 $v 2>&1 | sed -n 2,5p; done
