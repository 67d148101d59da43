set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/train_pmc
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d "$O/pmc_$c" -o p --output-format csv -- python "$R/tools/train_prof.py" 128 2 0 > "$O/$c.out" 2> "$O/$c.err"
done
ls $O/pmc_FETCH_SIZE | head
