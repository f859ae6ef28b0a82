#!/usr/bin/env python3
"""The 64x64-level self-attention launch a few times (PMC target): python scripts/attn_one.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.attn_microbench import run
run("self 64^2 d40", 16, 8, 40, 4096, 4096)
