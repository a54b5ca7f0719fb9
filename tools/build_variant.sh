#!/bin/bash
# builds build_abl/lib_<name>.so from build_abl/src/cnf_step3_<name>.hip + the current objects of the other sources
set -e
cd "$(dirname "$0")/../continuousnf.jl_amd/csrc"
for v in "$@"; do
  cp ../../build_abl/src/cnf_step3_$v.hip /tmp/cnf_step3_build_$v.hip   # (the .inc files it includes are found through -I.)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=fast -I. ${EXTRA:-} -c /tmp/cnf_step3_build_$v.hip -o ../../build_abl/cnf_step3_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_abl/lib_$v.so ../../build_abl/cnf_step3_$v.o cnf_mfma.o cnf_abi.o cnf_generic.o cnf_grad.o cnf_trace.o cnf_comm.o -ldl
  echo "built lib_$v.so"
done
