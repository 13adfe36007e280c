#!/bin/bash
# Scenes with participating media under several builds (development aid): usage dev_media_ab.sh <variant .so> ...
for i in 1 2; do
  for lib in "" "$@"; do
    echo "== $(basename "${lib:-current}")"
    RT_AMD_LIBRARY=$lib timeout -k 10 120 python scripts/dev_media.py || exit 1
  done
done
