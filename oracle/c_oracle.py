"""numpy binding of liboracle_c.so (the C half of the CPU oracle) -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_f32 = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_I = ctypes.c_int64


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle_c.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C {_HERE}`")
        _lib = ctypes.CDLL(path)
        _lib.orc_max_threads.restype = ctypes.c_int
    return _lib


def _p(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype).ctypes.data_as(ctypes.c_void_p)


def gather_rows(table, ids):
    table = np.ascontiguousarray(table, np.float32)
    ids = np.ascontiguousarray(ids, np.int64)
    out = np.empty(ids.shape + (table.shape[1],), np.float32)
    lib().orc_gather_rows_f32(_p(table, np.float32), _I(table.shape[0]), _I(table.shape[1]), _p(ids, np.int64),
                              _I(ids.size), out.ctypes.data_as(ctypes.c_void_p))
    return out


def gather_routed(item_table, user_table, ids, token_type):
    it = np.ascontiguousarray(item_table, np.float32)
    ut = np.ascontiguousarray(user_table, np.float32)
    ids = np.ascontiguousarray(ids, np.int64)
    tt = np.ascontiguousarray(token_type, np.int64)
    out = np.empty(ids.shape + (it.shape[1],), np.float32)
    lib().orc_gather_routed_f32(_p(it, np.float32), _I(it.shape[0]), _p(ut, np.float32), _I(ut.shape[0]),
                                _I(it.shape[1]), _p(ids, np.int64), _p(tt, np.int64), _I(ids.size),
                                out.ctypes.data_as(ctypes.c_void_p))
    return out


def scatter_add_rows(grad_rows, ids, n_rows, padding_idx=-1, token_type=None, want_type=0, out=None):
    d = grad_rows.shape[-1]
    g = np.ascontiguousarray(grad_rows, np.float32).reshape(-1, d)
    ids = np.ascontiguousarray(ids, np.int64).reshape(-1)
    tt = None if token_type is None else np.ascontiguousarray(token_type, np.int64).reshape(-1)
    if out is None:
        out = np.zeros((n_rows, d), np.float32)
    lib().orc_scatter_add_rows_f32(_p(g, np.float32), _p(ids, np.int64), _p(tt, np.int64), _I(want_type), _I(ids.size),
                                   _I(d), out.ctypes.data_as(ctypes.c_void_p), _I(n_rows), _I(padding_idx))
    return out


def score_topk(queries, table, k, exclude=None, id_offset=0):
    q = np.ascontiguousarray(queries, np.float32)
    t = np.ascontiguousarray(table, np.float32)
    ex = None if exclude is None else np.ascontiguousarray(exclude, np.int64)
    ids = np.empty((q.shape[0], k), np.int64)
    sc = np.empty((q.shape[0], k), np.float32)
    lib().orc_score_topk_f32(_p(q, np.float32), _I(q.shape[0]), _p(t, np.float32), _I(t.shape[0]), _I(t.shape[1]),
                             _I(k), _p(ex, np.int64), _I(0 if ex is None else ex.shape[1]), _I(id_offset),
                             ids.ctypes.data_as(ctypes.c_void_p), sc.ctypes.data_as(ctypes.c_void_p))
    return ids, sc


def topk_merge(cand_ids, cand_scores):
    ci = np.ascontiguousarray(cand_ids, np.int64)
    cs = np.ascontiguousarray(cand_scores, np.float32)
    S, nq, k = ci.shape
    ids = np.empty((nq, k), np.int64)
    sc = np.empty((nq, k), np.float32)
    lib().orc_topk_merge(_p(ci, np.int64), _p(cs, np.float32), _I(S), _I(nq), _I(k),
                         ids.ctypes.data_as(ctypes.c_void_p), sc.ctypes.data_as(ctypes.c_void_p))
    return ids, sc


def max_threads():
    return int(lib().orc_max_threads())
