"""Binary files of the reference's retrieval hand-off, and an exact top-K run over them.

Formats (little endian):
  * `save_emb(array, path)` (dataset.py:653-666, used for `embedding.fbin`, `query.fbin`, `id.u64bin` at
    embedding.py:681-684, infer.py:402): header = two uint32 `(n_rows, n_cols)`, then the array's raw bytes in
    row-major order.  `.fbin` holds float32 rows, `id.u64bin` holds uint64 ids as an (n, 1) array.
  * result file read by `read_result_ids` (infer.py:145-159): header = two uint32 `(n_queries, k)`, then
    `n_queries * k` uint64 ids, row-major.
`retrieve_files` plays the part of the external ANN step of infer.py:405-417: it reads the candidate vectors, the
candidate ids and the query vectors, scores every query against every candidate on the GPU (inner product, exact,
ties to the lower candidate row) and writes the result file that infer.py:420 reads next.
"""
import struct
from pathlib import Path

import numpy as np

_HEADER = struct.Struct('<II')


def save_emb(emb, save_path):
    """dataset.py:653-666: uint32 rows, uint32 cols, raw row-major bytes of `emb` (any dtype)."""
    emb = np.ascontiguousarray(emb)
    if emb.ndim != 2:
        raise ValueError("save_emb expects a 2-D array")
    with open(Path(save_path), 'wb') as f:
        f.write(_HEADER.pack(emb.shape[0], emb.shape[1]))
        emb.tofile(f)


def _load(path, dtype, mapped=False):
    """mapped: a copy-on-write memory map of the file instead of a read into a fresh array (the vectors then go
    from the page cache to the device without a second host copy)."""
    path = Path(path)
    with open(path, 'rb') as f:
        head = f.read(_HEADER.size)
        if len(head) != _HEADER.size:
            raise ValueError(f"{path}: truncated header")
        n, c = _HEADER.unpack(head)
        if mapped:
            have = (path.stat().st_size - _HEADER.size) // np.dtype(dtype).itemsize
            if have < n * c:
                raise ValueError(f"{path}: header says {n} x {c} {np.dtype(dtype).name}, file holds {have} values")
            if n * c == 0:
                return np.empty((n, c), dtype=dtype)
            return np.memmap(path, dtype=dtype, mode='c', offset=_HEADER.size, shape=(n, c))
        data = np.fromfile(f, dtype=dtype, count=n * c)
    if data.size != n * c:
        raise ValueError(f"{path}: header says {n} x {c} {np.dtype(dtype).name}, file holds {data.size} values")
    return data.reshape(n, c)


def load_fbin(path):
    """float32 (n, d) written by save_emb."""
    return _load(path, np.float32)


def load_u64bin(path):
    """uint64 ids written by save_emb as an (n, 1) array; returned flat."""
    return _load(path, np.uint64).reshape(-1)


def write_result_ids(ids, path):
    """The file layout `read_result_ids` (infer.py:145-159) expects: uint32 n_queries, uint32 k, uint64 ids."""
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    if ids.ndim != 2:
        raise ValueError("result ids must be (n_queries, k)")
    with open(Path(path), 'wb') as f:
        f.write(_HEADER.pack(ids.shape[0], ids.shape[1]))
        ids.tofile(f)


def read_result_ids(file_path):
    """infer.py:145-159."""
    return _load(file_path, np.uint64)


def retrieve_files(dataset_vector_file_path, dataset_id_file_path, query_vector_file_path, result_id_file_path,
                   query_ann_top_k=10, device='cuda', search=None):
    """Exact replacement of the ANN command of infer.py:405-417 (same four files, same k argument).

    search(queries float32 (Q,d) tensor, table float32 (N,d) tensor, k) -> int64 (Q,k) candidate ROW numbers;
    default: the HIP scoring + top-K on `device`.  Rows are mapped through the id file before writing, as the
    ANN tool reports dataset ids, not row numbers.  With fewer than k candidates the tail is filled with the
    last valid id of the row (the reader has no "missing" value)."""
    import torch
    hip = search is None
    table = _load(dataset_vector_file_path, np.float32, mapped=hip)
    ids = load_u64bin(dataset_id_file_path)
    queries = _load(query_vector_file_path, np.float32, mapped=hip)
    if ids.shape[0] != table.shape[0]:
        raise ValueError(f"{table.shape[0]} candidate vectors but {ids.shape[0]} ids")
    if queries.shape[1] != table.shape[1]:
        raise ValueError(f"query dim {queries.shape[1]} != candidate dim {table.shape[1]}")
    k = int(query_ann_top_k)
    kk = min(k, table.shape[0])
    if hip:
        # device path: vectors straight from the file mapping, the row -> id mapping as a device gather (the numpy
        # form of that step alone took 73 ms for 1 M x 10 rows), one copy of the finished ids back
        from . import ops
        t_t = torch.from_numpy(table).to(device)
        q_t = torch.from_numpy(queries).to(device)
        rows_t = ops.score_topk(q_t, ops.PreparedTable(t_t), kk)[0]
        if kk < k:
            rows_t = torch.cat([rows_t, rows_t[:, -1:].expand(-1, k - kk)], dim=1)
        rows_t = rows_t.clamp_min(0)
        ids_t = torch.from_numpy(ids.view(np.int64)).to(device)          # uint64 bit patterns carried as int64
        out = ids_t[rows_t].cpu().numpy().view(np.uint64)
        write_result_ids(out, result_id_file_path)
        return rows_t.cpu().numpy()
    rows = search(torch.from_numpy(queries), torch.from_numpy(table), kk).cpu().numpy()
    if kk < k:
        rows = np.concatenate([rows, np.repeat(rows[:, -1:], k - kk, axis=1)], axis=1)
    rows = np.where(rows < 0, 0, rows)
    write_result_ids(ids[rows], result_id_file_path)
    return rows
