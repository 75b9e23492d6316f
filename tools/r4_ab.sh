#!/bin/bash
# usage: VARIANTS="...;..." bash tools/r4_ab.sh   (A/B on minimal shapes, then restores nothing: local build is unaffected)
bash tools/ab_rows4.sh
