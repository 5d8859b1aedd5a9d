# A/B/A/B of two builds of the library on one box: ab_lib.sh <variant .so name> <probe command ...>
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
V=$1; shift
for i in 1 2; do
  echo "== default"; "$@"
  echo "== $V"; SATCV_LIB=$R/satellite_computervision_amd/$V "$@"
done
