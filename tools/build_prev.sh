#!/bin/bash
# Build the library of another commit next to the real one, for same-box A/B runs:
#   tools/build_prev.sh [rev = HEAD]  ->  tools/abl/prev.so ; run anything against it with FH_LIB_PATH=...
# (sources of that revision are checked out to a scratch directory under /tmp; the working tree is not touched)
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}
d=/tmp/fh_prev_src; rm -rf $d; mkdir -p $d/flowhigh_amd $d/include tools/abl
git archive $rev flowhigh_amd/csrc include | tar -x -C $d
objs=""
for f in $d/flowhigh_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -c $f -o $f.o &
  objs="$objs $f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/abl/prev.so $objs
echo tools/abl/prev.so "($rev)"
