set -e
cd $GRAFT_REPO_ROOT
for wp in 3 2; do
  export BP_MSM_WPIPE=$wp
  echo "=== WPIPE $wp"
  bash scripts/msm_timeline.sh "5,4" 2>&1 | grep -v "^\[" | tail -95
done
