#!/bin/bash
python tools/mb_lanczos.py 2>&1 | grep -v "^    " | tail -6
