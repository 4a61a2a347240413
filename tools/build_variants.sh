# builds compile-time variants of the library for A/B runs: tools/variants/libilqr_<name>.so
#   usage: tools/build_variants.sh name1 "-DFLAG=.." name2 "-DFLAG=.." ...
set -e
cd "$(dirname "$0")/../iterative-linear-quadratic-regulator_amd/csrc"
mkdir -p ../../tools/variants
build() { # name flags...
  name=$1; shift
  for f in ilqr_f32 ilqr_f64; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-value -Wno-unused-result -ffp-contract=on -mllvm -amdgpu-mfma-vgpr-form -mllvm -slp-threshold=2 $@ -c -o /tmp/v_${name}_$f.o $f.hip & done
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-value -x hip -c -o /tmp/v_${name}_abi.o ilqr_abi.cpp &
  wait
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/variants/libilqr_$name.so /tmp/v_${name}_ilqr_f32.o /tmp/v_${name}_ilqr_f64.o /tmp/v_${name}_abi.o -ldl
}
while [ $# -ge 2 ]; do build "$1" $2; shift 2; done
ls -la ../../tools/variants
