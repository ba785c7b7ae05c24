#!/bin/bash
# A/B of a variant build of the library against the default one, in ONE gpurun call.
# usage: tools/ab_variant.sh <variant.so> <command...>   (env BJX_* passes through)
V=$1; shift
cp blackjax_amd/libbjxhip.so /tmp/libbjxhip_default.so
echo "== default"; "$@"
cp $V blackjax_amd/libbjxhip.so
echo "== variant $V"; "$@"
cp /tmp/libbjxhip_default.so blackjax_amd/libbjxhip.so
