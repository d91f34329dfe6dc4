"""Small host utilities of the data path (reference: cubercnn/util/util.py:15-32,121)."""
import json
import os
import shutil


def file_parts(file_path):
    """('dir', 'name', '.ext')"""
    folder, tail = os.path.split(file_path)
    name, ext = os.path.splitext(tail)
    return folder, name, ext


def save_json(path, data):
    with open(path, 'w') as f:
        json.dump(data, f)


def load_json(path):
    with open(path, 'r') as f:
        return json.load(f)


def mkdir_if_missing(directory, delete_if_exist=False):
    if delete_if_exist and os.path.exists(directory):
        shutil.rmtree(directory)
    os.makedirs(directory, exist_ok=True)
