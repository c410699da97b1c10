#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>: an A/B build of libbrmi.so under scratch/variants/ (loaded through BRMI_LIB_PATH)
set -e
NAME=$1; shift
mkdir -p scratch/variants
make -s -j8 hip EXTRA="$*" LIBDIR=scratch/variants/$NAME OBJDIR=build/variants/$NAME >/dev/null
echo scratch/variants/$NAME/libbrmi.so
