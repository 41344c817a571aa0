# Run ON THE GPU BOX: tools/wino_stamps.sh for several flag sets ("name:flags" ...)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  echo "=== $name ($flags)"
  EXTRA="$flags" bash $(dirname $0)/wino_stamps.sh 2>&1 | grep -v amdgpu.ids
done
