#!/bin/bash
python tools/mb_dist_overlap.py 2>&1 | grep -E "^reserve"
