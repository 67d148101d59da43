run() { echo "== $*"; env "$@" timeout 60 python tools/train_prof.py 128 10 block 2>&1 | sed -n 2,5p; }
run FEAR_LIN_ROWS=100000 FEAR_LIN_CIN=32
run FEAR_LIN_ROWS=0 FEAR_LIN_CIN=32
run FEAR_LIN_ROWS=0 FEAR_LIN_CIN=64
run FEAR_LIN_ROWS=0 FEAR_LIN_CIN=128
run FEAR_LIN_ROWS=1000000000 FEAR_LIN_CIN=32
