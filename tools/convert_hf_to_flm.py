#!/usr/bin/env python3
"""HF LLaMA checkpoint directory -> .flm (see fast-llama_amd/convert.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
from fast_llama_amd import convert
sys.exit(convert.main())
