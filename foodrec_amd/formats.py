"""On-disk formats either side of the scoring path, plus synthetic generators in those formats.

Formats (reference: ``Code/Recommender/Dataset.py:20-71``, ``Train_recommender.py:99-106, :124-133``):

* ``<base>.train.rating`` / ``<base>.test.rating`` -- one ``user\\titem[\\t...]`` line per interaction;
  readers keep the first two integer fields and group items per user under the key ``str(user)``.
* ``<base>.test.negative`` -- ``<c>user<c>\\tneg\\tneg...``: the first field is the user id wrapped in one
  character on each side (stripped positionally, ``Dataset.py:45-47``), then >= 100 negative dish ids
  (the first 50 feed training, ``Train_recommender.py:86-87``; ``[50:100]`` feed evaluation,
  ``evaluate.py:45``).
* ``Personal_Memory.npy [U, C+1, E]``, ``Recipe_Embedding.npy [I, E]``, ``Category_Embedding.npy [C, E]``,
  ``General_Memory.npy [L, C+1, E]`` -- float32.
* ``dish_to_category.json`` ``{str(dish): [[m0], ..., [mC-1]]}``; ``user_to_one_hot_label.json``
  ``{str(user): [L floats]}``.

The real FoodRec split is not redistributable/offline; `write_synthetic_split` produces files of the
same shape so the whole plumbing (BASELINE.json config 1) can run.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List

import numpy as np


def _grouped_pairs(path: str) -> Dict[str, List[int]]:
    out: Dict[str, List[int]] = {}
    with open(path, "r") as f:
        for line in f:
            if line == "":
                break
            fields = line.split("\t")
            user, item = int(fields[0]), int(fields[1])
            out.setdefault(str(user), []).append(item)
    return out


def load_rating_file_as_list(path: str) -> Dict[str, List[int]]:
    """``Dataset.load_rating_file_as_list`` (Dataset.py:20-36)."""
    return _grouped_pairs(path)


def load_rating_file_as_matrix(path: str) -> Dict[str, List[int]]:
    """``Dataset.load_rating_file_as_matrix`` (Dataset.py:55-71) -- same grouping, train file."""
    return _grouped_pairs(path)


def load_negative_file(path: str) -> Dict[str, List[int]]:
    """``Dataset.load_negative_file`` (Dataset.py:38-53): key = first field minus its first and last character."""
    out: Dict[str, List[int]] = {}
    with open(path, "r") as f:
        for line in f:
            if line == "":
                break
            fields = [x.strip("\n") for x in line.split("\t")]
            key = fields[0][1:-1]
            out[key] = [int(x) for x in fields[1:]]
    return out


class Dataset:
    """``Dataset(path)`` (Dataset.py:3-18): the three dicts and the three counts."""

    def __init__(self, path: str):
        self.trainMatrix = load_rating_file_as_matrix(path + ".train.rating")
        self.testRatings = load_rating_file_as_list(path + ".test.rating")
        self.testNegatives = load_negative_file(path + ".test.negative")
        self.num_train_users = len(self.trainMatrix)
        self.num_instances = sum(len(v) for v in self.trainMatrix.values())
        self.num_test = sum(len(v) for v in self.testRatings.values())


def get_train_instances(train: Dict[str, List[int]], testNegatives: Dict[str, List[int]], dish_to_category: Dict[str, list],
                        user_to_one_hot_label: Dict[str, list], rng=None):
    """The training feeds the reference's driver builds once per run (``Train_recommender.py:69-93``): per user of
    ``train`` (dict order), up to 200 of its positives drawn with ``random.sample`` (label 1, write_sign [1.0]), then
    the first 50 of its test negatives (label 0, write_sign [-1.0]); ``categories`` and ``user_one_hot_label`` are
    looked up per instance.  Returns the six parallel lists in the reference's order:
    ``user_input_index, item_input_index, labels, categories, write_sign, user_one_hot_label``.

    ``rng`` is a ``random.Random`` (default: the ``random`` module itself, as in the reference, so seeding
    ``random`` reproduces its draw)."""
    import random as _random
    rng = rng or _random
    users, items, labels, categories, write_sign, one_hot = [], [], [], [], [], []
    for user in train:
        positives = train[str(user)]
        for dish in rng.sample(positives, min(len(positives), 200)):           # :72-73
            users.append(user); items.append(dish); labels.append(1)
            categories.append(dish_to_category[str(dish)])
            write_sign.append([1.0])
            one_hot.append(user_to_one_hot_label[str(user)])
        for dish in testNegatives[str(user)][:50]:                                # :81-82
            users.append(user); items.append(dish); labels.append(0)
            categories.append(dish_to_category[str(dish)])
            write_sign.append([-1.0])
            one_hot.append(user_to_one_hot_label[str(user)])
    return users, items, labels, categories, write_sign, one_hot


def load_numpy_file(path: str) -> np.ndarray:
    return np.load(path)                      # Train_recommender.py:99-101


def load_json_file(path: str) -> dict:
    with open(path, "r") as f:                # Train_recommender.py:104-106
        return json.loads(f.read())


# ---- synthetic data in the reference's formats ------------------------------------------------------

def synthetic_tables(num_users: int, num_dishes: int, num_categories: int, embed_size: int, num_labels: int = 95,
                     seed: int = 20260101):
    """N(0, 1/E) float32 tables and a random non-empty category subset per dish."""
    rng = np.random.default_rng(seed)
    s = np.float32(1.0 / np.sqrt(embed_size))
    pm = rng.standard_normal((num_users, num_categories + 1, embed_size), dtype=np.float32) * s
    re = rng.standard_normal((num_dishes, embed_size), dtype=np.float32) * s
    ce = rng.standard_normal((num_categories, embed_size), dtype=np.float32) * s
    gm = rng.standard_normal((num_labels, num_categories + 1, embed_size), dtype=np.float32) * s
    pattern = rng.integers(1, 2 ** num_categories, num_dishes)          # non-empty subset
    cats = ((pattern[:, None] >> np.arange(num_categories)[None, :]) & 1).astype(np.float32)
    return pm, re, ce, gm, cats


def write_synthetic_split(directory: str, dataset: str = "foodrec-synth", num_users: int = 64657,
                          num_dishes: int = 4548, num_categories: int = 4, num_labels: int = 95,
                          embed_size: int = 32, num_test_users: int = 0, train_per_user: int = 3,
                          num_negatives: int = 100, seed: int = 20260101) -> str:
    """Write every file ``Train_recommender.py`` loads (:124-133) at the reference's default sizes
    (:51-58).  Returns the ``path + dataset`` prefix to give ``Dataset``."""
    os.makedirs(directory, exist_ok=True)
    rng = np.random.default_rng(seed + 1)
    pm, re, ce, gm, cats = synthetic_tables(num_users, num_dishes, num_categories, embed_size, num_labels, seed)
    np.save(os.path.join(directory, "Personal_Memory.npy"), pm)
    np.save(os.path.join(directory, "Recipe_Embedding.npy"), re)
    np.save(os.path.join(directory, "Category_Embedding.npy"), ce)
    np.save(os.path.join(directory, "General_Memory.npy"), gm)
    with open(os.path.join(directory, "dish_to_category.json"), "w") as f:
        json.dump({str(d): [[float(x)] for x in cats[d]] for d in range(num_dishes)}, f)
    labels = (rng.random((num_users, num_labels)) < 0.03).astype(np.float32)
    labels[labels.sum(1) == 0, 0] = 1.0
    with open(os.path.join(directory, "user_to_one_hot_label.json"), "w") as f:
        json.dump({str(u): [float(x) for x in labels[u]] for u in range(num_users)}, f)
    n_test = num_users if num_test_users <= 0 else min(num_test_users, num_users)
    base = os.path.join(directory, dataset)
    with open(base + ".train.rating", "w") as ftr, open(base + ".test.rating", "w") as fte, \
            open(base + ".test.negative", "w") as fng:
        for u in range(n_test):
            for d in rng.integers(0, num_dishes, train_per_user):
                ftr.write("%d\t%d\t1\t0\n" % (u, d))
            pos = int(rng.integers(0, num_dishes))
            fte.write("%d\t%d\t1\t0\n" % (u, pos))
            negs = rng.integers(0, num_dishes, num_negatives)
            fng.write("(%d)\t%s\n" % (u, "\t".join(str(int(x)) for x in negs)))
    return base
