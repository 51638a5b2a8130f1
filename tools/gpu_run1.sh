#!/bin/bash
python tools/mb_iql_profile.py 2>&1 | grep -v "^\[W\|amdgpu.ids" | cut -c1-180 | head -50
