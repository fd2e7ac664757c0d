#!/usr/bin/env python3
"""tools/tuner_value_probe.py -- what the online tuner's decision is worth on the SAME handle (same state block), sustained: every leg calls until
the tuner has decided (cvs_launch_info.tune_state), then tuner-on and tuner-off (CVS_OPT_AUTOTUNE 0 = the engine's default configuration) take
turns, 3 x 150 launches each.  Legs: full setup, caller pipeline, G4, a mid-size full setup, and -- since round 6, permanently -- the two 32 x 1080p
batch kinds of BASELINE config 4.  Run it in many processes (the tuner's memory is per process): a kept challenger that is more than 1 % behind the
default sustained is a wrong decision.  (The legs live in tools/r06_probe.py, section `tune`.)"""
import os, runpy, sys
sys.argv = [sys.argv[0], "tune"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "r06_probe.py"), run_name="__main__")
