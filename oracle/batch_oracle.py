"""CPU oracle for device-side batch assembly.  TEST INFRASTRUCTURE ONLY (only tests/ may import it).

Restates, in numpy, what the reference does per sample in ``Dataset.__getitem__``
(/root/reference/fuxictr/pytorch/data_generator.py:66-78) followed by torch's default collate (:239-241) and the slicing
at the top of ``RAT_m2.forward`` (/root/reference/fuxictr/pytorch/models/RAT_m2.py:110-116).

Parity status: PINNED — ``tests/golden/make_golden_batch.py`` runs the real reference ``Dataset`` under a ``DataLoader``
and commits the collated batches (``tests/golden/batch_assembly.npz``); ``tests/test_batch_assembly.py`` checks
``assemble_batch`` against them before using it to check the HIP kernel.
"""
import numpy as np


def assemble_batch(data, pool, retr_indices, rows):
    """data [Q, L+1] / pool [N, L+1] (label last), retr_indices [Q, K], rows [B] -> (X [B,1+K,L], y [B,1+K]) float64.

    data_generator.py:67  darray_i = self.darray[index]
    data_generator.py:69  retrieved = self.retr_pool_darray[self.retr_indices[index]]   (numpy fancy index: -1 = last row)
    data_generator.py:70-71  concat([darray_i[None], retrieved])
    data_generator.py:72-73  X_i = darray_i[..., :-1]; y_i = darray_i[..., -1]
    """
    data, pool = np.asarray(data, dtype=np.float64), np.asarray(pool, dtype=np.float64)
    X, y = [], []
    for r in rows:
        sample = np.concatenate([data[r][None], pool[retr_indices[r]]])
        X.append(sample[..., :-1])
        y.append(sample[..., -1])
    return np.stack(X), np.stack(y)


def model_inputs(X, y):
    """What the model consumes: int32 ids, label-token ids (target row -> 2, RAT_m2.py:116), fp32 target labels."""
    idx = X.astype(np.int32)
    label_ids = y.astype(np.int32).copy()
    label_ids[:, 0] = 2
    return idx, label_ids, y[:, 0].astype(np.float32)
