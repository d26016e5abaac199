# same-box A/B of the headline launch: tools/scratch/ab_headline.sh <variant.so> [reps]
V=$1; R=${2:-4}
cd $GRAFT_REPO_ROOT
for i in $(seq $R); do
  unset SVAE_AMD_LIB; echo -n "default: "; python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['achieved'])"
  export SVAE_AMD_LIB=$GRAFT_REPO_ROOT/$V; echo -n "variant: "; python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['achieved'])"
done
