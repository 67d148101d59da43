run() { echo "== $*"; env "$@" timeout 60 python tools/train_prof.py 128 10 block 2>&1 | sed -n 2,6p; }
run FEAR_DBG_SKIP=32
run FEAR_DBG_SKIP=0
run FEAR_DBG_SKIP=32
