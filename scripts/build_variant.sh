#!/bin/bash
# Build an experimental variant of libsvcmi.so with extra -D flags: scripts/build_variant.sh <name> <flags...>  -> whisper-vits-svc_amd/svcmi/exp/libsvcmi_<name>.so
# (SVCMI_LIB=<path> makes svcmi load it; A/B measurements on the GPU box without rebuilding there)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../whisper-vits-svc_amd"
mkdir -p _obj/$NAME svcmi/exp
rm -f _obj/$NAME/*.o svcmi/exp/libsvcmi_$NAME.so
for f in csrc/*.hip; do
  EXTRA=""
  [ "$(basename $f)" = amp_fused.hip ] && [ -z "$SVCMI_VARIANT_AMP_ACC_FILE" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1"      # (build.py FILE_FLAGS)
  [ "$(basename $f)" = conv_gemm.hip ] && [ -z "$SVCMI_VARIANT_ACC_FILE" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1 -DSVCMI_ACC_IN_VGPRS=1 -DSVCMI_DMA_M0_RAW=1"      # (SVCMI_VARIANT_ACC_FILE=1: accumulator-file form, the build before the pinned K loop)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EXTRA "$@" -c $f -o _obj/$NAME/$(basename $f).o &
done
wait
for f in csrc/*.hip; do [ -s _obj/$NAME/$(basename $f).o ] || { echo "build_variant: $f did not compile" >&2; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o svcmi/exp/libsvcmi_$NAME.so _obj/$NAME/*.o
echo svcmi/exp/libsvcmi_$NAME.so
