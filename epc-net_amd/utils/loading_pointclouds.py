"""Host-side data path of the reference (``utils/loading_pointclouds.py``) plus the hard-negative selection of
``train.py``.  File I/O and tuple sampling stay on the host (numpy / pickle) exactly as in the reference; the one
compute step -- nearest latent vectors for hard-negative mining, sklearn KDTree in ``train.py:857-869`` -- runs on the
GPU (``epc_pairwise_topk``).

Formats: ``*.bin`` = 4096 x 3 float64 little-endian (``load_pc_file`` :26-53); training pickle = {key: {'query': file,
'positives': [keys], 'negatives': [keys]}} (:11-16); evaluation pickle = [ {key: {'query', 'northing', 'easting',
<run index>: [true neighbour keys]}} per run ] (:18-24, generate_test_sets.py:88-104).
"""
from __future__ import annotations

import os
import pickle
import random
from typing import Dict, List, Optional, Sequence

import numpy as np

NUM_POINTS = 4096


def get_queries_dict(filename: str) -> dict:
    """:11-16."""
    with open(filename, "rb") as handle:
        return pickle.load(handle)


def get_sets_dict(filename: str) -> list:
    """:18-24."""
    with open(filename, "rb") as handle:
        return pickle.load(handle)


def load_pc_file(filename: str, dataset_folder: str, input_dim: int = 3) -> np.ndarray:
    """:26-53.  A file of the wrong length yields an all-zero cloud (the reference prints and returns zeros too)."""
    raw = np.fromfile(os.path.join(dataset_folder, filename), dtype=np.float64)
    if raw.shape[0] != NUM_POINTS * input_dim:
        print("Error in pointcloud shape", raw.shape, filename)
        return np.zeros([NUM_POINTS, input_dim])
    pc = raw.reshape(NUM_POINTS, input_dim)
    if input_dim != 3:
        # 13-d variant (:40-51): min-max normalise the hand-crafted columns 3..11, NaN -> 0, inf -> 1
        span = pc.max(axis=0) - pc.min(axis=0)
        with np.errstate(divide="ignore", invalid="ignore"):
            pc[:, 3:12] = ((pc - pc.min(axis=0)) / span)[:, 3:12]
        pc[np.isnan(pc)] = 0.0
        pc[np.isinf(pc)] = 1.0
    return pc


def load_pc_files(filenames: Sequence[str], dataset_folder: str, input_dim: int = 3) -> np.ndarray:
    """:56-65."""
    clouds = [load_pc_file(f, dataset_folder, input_dim) for f in filenames]
    return np.array([c for c in clouds if c.shape[0] == NUM_POINTS])


def load_pc_data(entries: Dict[int, dict], dataset_folder: str, input_dim: int = 3) -> np.ndarray:
    """train.py:159-190: every cloud of a query dict into one (T, 4096, D) float32 array, in key order."""
    names = [entries[i]["query"] for i in range(len(entries))]
    return load_pc_files(names, dataset_folder, input_dim).astype(np.float32)


def rotate_point_cloud(batch_data: np.ndarray) -> np.ndarray:
    """:67-88 (unused by the reference's training loop: call sites commented out, train.py:393-398)."""
    out = np.zeros(batch_data.shape, dtype=np.float32)
    for k in range(batch_data.shape[0]):
        ang = np.random.uniform() * np.pi - np.pi / 2.0
        c, s = np.cos(ang), np.sin(ang)
        rot = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        out[k] = batch_data[k].reshape(-1, 3) @ rot
    return out


def jitter_point_cloud(batch_data: np.ndarray, sigma: float = 0.005, clip: float = 0.05) -> np.ndarray:
    """:90-100."""
    assert clip > 0
    return batch_data + np.clip(sigma * np.random.randn(*batch_data.shape), -clip, clip)


def get_query_tuple(idx, dict_value, num_pos, num_neg, QUERY_DICT, hard_neg=[], other_neg=False, dataset_folder=None,
                    data=None):
    """:102-168.  [query, positives, negatives(, other negative)] drawn from the preloaded ``data`` array; hard
    negatives come first, the rest is filled from the shuffled negative list; the "other negative" of the quadruplet
    loss is a cloud that is a positive of neither the query nor any chosen negative (empty array if none exists)."""
    query = data[idx]
    random.shuffle(dict_value["positives"])
    positives = data[[dict_value["positives"][i] for i in range(num_pos)]]
    random.shuffle(dict_value["negatives"])
    if len(hard_neg) == 0:
        neg_indices = [dict_value["negatives"][i] for i in range(num_neg)]
    else:
        neg_indices = list(hard_neg)
        j = 0
        while len(neg_indices) < num_neg:
            if dict_value["negatives"][j] not in hard_neg:
                neg_indices.append(dict_value["negatives"][j])
            j += 1
    negatives = data[neg_indices]
    if not other_neg:
        return [query, positives, negatives]
    neighbors = list(dict_value["positives"])
    for neg in neg_indices:
        neighbors.extend(QUERY_DICT[neg]["positives"])
    possible_negs = list(set(QUERY_DICT.keys()) - set(neighbors))
    random.shuffle(possible_negs)
    if len(possible_negs) == 0:
        return [query, positives, negatives, np.array([])]
    return [query, positives, negatives, data[possible_negs[0]]]


def get_random_hard_negatives(query_vec, random_negs: Sequence[int], num_to_take: int, latent_vectors,
                              search=None) -> List[int]:
    """train.py:857-869: the ``num_to_take`` sampled negatives whose cached descriptors are nearest to the query's.
    ``search(database, queries, k) -> (dist, idx)`` defaults to the GPU exact k-NN (retrieval.knn_search)."""
    import torch
    if search is None:
        from ..retrieval import knn_search as search
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    lat = torch.as_tensor(np.asarray(latent_vectors)[list(random_negs)], dtype=torch.float32, device=dev)
    q = torch.as_tensor(np.asarray(query_vec), dtype=torch.float32, device=dev).reshape(1, -1)
    _, idx = search(lat, q, num_to_take)
    return np.asarray(random_negs)[idx[0].cpu().numpy()].tolist()
