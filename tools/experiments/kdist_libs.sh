# bash tools/experiments/kdist_libs.sh <pattern> lib1.so lib2.so ...: kernel duration distribution per library build (training bench)
PAT=$1; shift
for l in "$@"; do echo "== $l"; IODINE_HIP_LIB=${GRAFT_REPO_ROOT:-$(pwd)}/$l bash tools/experiments/kdist.sh "$PAT" | cut -c1-200; done
