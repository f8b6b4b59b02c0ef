#!/bin/bash
(timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8)
bash tools/ab_e.sh new
AB_CONFIG=B bash tools/ab_e.sh new
