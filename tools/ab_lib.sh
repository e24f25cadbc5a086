# developer A/B (GPU box): two builds of the library back to back; usage: bash tools/ab_lib.sh <old.so> "<command>"
OLD=$1; shift
for i in 1 2; do
  echo "--- new"; "$@"
  echo "--- old ($OLD)"; OCTIC_LIB=$OLD "$@"
done
