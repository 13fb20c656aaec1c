cd /tmp && export TMPDIR=/tmp
for cfg in "0 8" "0 1" "0 4" "2 8"; do set -- $cfg
echo "== HNR_MARCH_PROBE=$1 rays_per_wave=$2"; HNR_MARCH_PROBE=$1 HNR_MARCH_RAYS_PER_WAVE=$2 PROBE_PAD=0 HNR_KNN=4 PROBE_KNN_ORDER=1 timeout 600 python3 $GRAFT_REPO_ROOT/tools/probe_query.py 2>&1 | grep -E "march\+knn"
done
