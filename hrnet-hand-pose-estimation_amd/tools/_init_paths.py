"""put <package>/lib on sys.path (same idiom as the reference's tools/_init_paths.py)"""
import os.path as osp
import sys

this_dir = osp.dirname(osp.abspath(__file__))
lib_path = osp.join(this_dir, '..', 'lib')
if lib_path not in sys.path:
    sys.path.insert(0, lib_path)
