#!/bin/bash
python tools/mb_kron_rate.py 2>&1 | tail -10
