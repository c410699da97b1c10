"""Test-side bridge to the reference's own cluster-LOD builder (oracle/_ref/libclodref.so).

oracle/_ref/libclodref.so is the reference's clusterlod.h + vendored meshoptimizer, compiled by oracle/ref/Makefile from the
sources where they lie under /root/reference.  It is TEST INFRASTRUCTURE: only tests load it, through this module, and hand its
`clodref_dag_build` / `clodref_dag_release` entry points to libbrmi_scene.so as a caller-supplied DAG builder
(brmi_scene_create_with_dag_builder).  Nothing under basicrenderer_amd/ and nothing in bench.py's measured path knows about it.
"""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libclodref.so")
_lib = None


def available():
    return os.path.exists(LIB)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError(f"{LIB} is missing: make -C oracle/ref (needs /root/reference)")
        _lib = C.CDLL(LIB)
    return _lib


def dag_builder():
    """(build_fn, release_fn) addresses for Scene(dag_builder=...)."""
    l = lib()
    return (C.cast(l.clodref_dag_build, C.c_void_p), C.cast(l.clodref_dag_release, C.c_void_p))
