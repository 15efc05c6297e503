#!/bin/bash
# Builds a VARIANT of libfdmi.so next to the in-tree one for library-level A/B runs (FDMI_LIB=<path> selects it):
#   scripts/build_variant.sh NAME "-DFLAG ..." file1.hip file2.hip ...
# compiles the named translation units with the extra flags into build/variants/NAME/ and links them with the in-tree objects of the
# other units (run `make -C flash_diffusion_amd/csrc` first) -> build/variants/NAME/libfdmi.so  (build/ is git-ignored, travels to the box)
set -eu
name=$1; flags=$2; shift 2
root="$(cd "$(dirname "$0")/.." && pwd)"
src="$root/flash_diffusion_amd/csrc"
dst="$root/build/variants/$name"
mkdir -p "$dst"
objs=""
for o in "$src"/*.o; do
  b=$(basename "$o" .o)
  case " $* " in *" $b.hip "*) ;; *) objs="$objs $o";; esac
done
for f in "$@"; do
  b=$(basename "$f" .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -c "$src/$f" -o "$dst/$b.o" &
done
wait
for f in "$@"; do objs="$objs $dst/$(basename "$f" .hip).o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$dst/libfdmi.so" $objs -ldl
echo "$dst/libfdmi.so"
