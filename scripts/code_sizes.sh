#!/bin/bash
# Code size (bytes) of every gfx950 kernel of one translation unit -- to keep the hot loops inside the 64 KB instruction cache.
#   bash scripts/code_sizes.sh torchcde_amd/csrc/dopri5_mlp_adjoint.hip [min_bytes]
SRC=$(readlink -f "$1"); MIN=${2:-0}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/cde_sizes_XXXX)
EXTRA=$(cd $ROOT && python -c "
from torchcde_amd import _lib
import sys, os
print(' '.join(_lib.EXTRA_FLAGS.get(os.path.basename('$SRC'), [])))")
(cd $TMP && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $EXTRA -c $SRC -o x.o --save-temps >/dev/null 2>&1
 /opt/rocm/lib/llvm/bin/llvm-readelf -s --wide *gfx950*.o | grep FUNC | awk -v m=$MIN '$3>=m {print $3, $8}' | c++filt | sort -n | sed 's/(float const\*.*//; s/(cde::.*//')
rm -rf $TMP
