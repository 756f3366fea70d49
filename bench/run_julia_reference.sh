#!/bin/bash
# Runs bench/julia_reference.jl when a julia binary exists; this image has none.
if command -v julia >/dev/null 2>&1; then
  exec julia "$(dirname "$0")/julia_reference.jl" "$@"
else
  echo "SKIPPED: julia not found"
fi
