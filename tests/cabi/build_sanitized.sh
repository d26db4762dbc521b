#!/bin/bash
# usage: build_sanitized.sh address|thread OUTDIR — TEST INFRASTRUCTURE: the HOST halves of csrc/*.hip (hipcc --cuda-host-only), the HIP stand-in
# (hip_host_stub.cpp), the librccl stand-in (fake_rccl.cpp) and the driver (sanitize_host.cpp), all instrumented with one sanitizer.  CPU container, no GPU.
set -e
SAN=$1; OUT=$2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CXX=/opt/rocm/lib/llvm/bin/clang++
mkdir -p $OUT
FL="--cuda-host-only -std=c++17 -O1 -g -fPIC -fsanitize=$SAN -fno-omit-frame-pointer -ffp-contract=off -Wno-unused-value -Wno-unused-result -Wno-pass-failed -I$ROOT/include"
pids=()
for f in $ROOT/meepoembedding_amd/csrc/*.hip; do
  o=$OUT/$(basename $f .hip).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find $ROOT/meepoembedding_amd/csrc -name '*.h' -newer $o)" ]; then $HIPCC $FL -c $f -o $o & pids+=($!); fi
done
$HIPCC $FL -x hip -c $ROOT/tests/cabi/hip_host_stub.cpp -o $OUT/hip_host_stub.o & pids+=($!)
$HIPCC $FL -x hip -c $ROOT/tests/cabi/fake_rccl.cpp -o $OUT/fake_rccl.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
# the host-only objects refer to the fat binary hipcc would have embedded: give every such symbol a body nobody reads
nm -u $OUT/*.o | grep -o "__hip_fatbin_[0-9a-f]*" | sort -u | awk '{print "char "$1"[8];"}' > $OUT/fatbin_syms.c
gcc -c -fPIC $OUT/fatbin_syms.c -o $OUT/fatbin_syms.o
$CXX -shared -fsanitize=$SAN $OUT/meepo_*.o $OUT/hip_host_stub.o $OUT/fatbin_syms.o -ldl -lpthread -o $OUT/libmeepo_host.so
$CXX -shared -fsanitize=$SAN $OUT/fake_rccl.o -L$OUT -lmeepo_host -lrt -Wl,-rpath,$OUT -o $OUT/libfake_rccl.so
$CXX -std=c++17 -O1 -g -fsanitize=$SAN -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/include $ROOT/tests/cabi/sanitize_host.cpp -L$OUT -lmeepo_host -lpthread -Wl,-rpath,$OUT -o $OUT/sanitize_host
echo "built $OUT/sanitize_host ($SAN)"
