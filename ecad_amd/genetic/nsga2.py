"""Minimal NSGA-II with the reference's operator settings (``ecad/genetic/train_nsga2_base.py:234-242``):
population 72, ``PointCrossover(prob=0.9, n_points=4)``, ``BitflipMutation(prob=0.05)``, seed 0, ask/tell with offline evaluation.

Mutation follows pymoo 0.6.1.3's semantics [ext]: ``prob`` is the chance that an offspring is mutated AT ALL; a mutated offspring flips
each bit with ``prob_var``, which pymoo defaults to ``min(0.5, 1 / n_var)`` (1/1680 here) -- about 0.05 expected flips per child, not 84.

Stand-in for ``pymoo==0.6.1.3`` (``requirements.txt:39``), which is not installed here: same algorithm family
(constraint-dominated fast non-dominated sort, crowding distance, binary tournament, duplicate elimination) and
hyper-parameters, NOT pymoo's RNG stream -- populations are not bit-identical to a pymoo run.  The GA operators are
CPU microseconds and outside the hot-path scope; this exists so the sharded evaluation loop can be driven end to end.
"""
from __future__ import annotations

import numpy as np


def dominates(f1, g1, f2, g2) -> bool:
    cv1, cv2 = max(g1, 0.0), max(g2, 0.0)
    if cv1 != cv2:
        return cv1 < cv2
    return bool(np.all(f1 <= f2) and np.any(f1 < f2))


def fast_non_dominated_sort(F: np.ndarray, G: np.ndarray) -> list[list[int]]:
    n = len(F)
    S, cnt, fronts = [[] for _ in range(n)], np.zeros(n, dtype=int), [[]]
    for p in range(n):
        for q in range(n):
            if p == q:
                continue
            if dominates(F[p], G[p], F[q], G[q]):
                S[p].append(q)
            elif dominates(F[q], G[q], F[p], G[p]):
                cnt[p] += 1
        if cnt[p] == 0:
            fronts[0].append(p)
    i = 0
    while fronts[i]:
        nxt = []
        for p in fronts[i]:
            for q in S[p]:
                cnt[q] -= 1
                if cnt[q] == 0:
                    nxt.append(q)
        fronts.append(nxt)
        i += 1
    return fronts[:-1]


def crowding_distance(F: np.ndarray) -> np.ndarray:
    n, m = F.shape
    d = np.zeros(n)
    for k in range(m):
        order = np.argsort(F[:, k], kind="stable")
        span = F[order[-1], k] - F[order[0], k]
        d[order[0]] = d[order[-1]] = np.inf
        if span > 0 and n > 2:
            d[order[1:-1]] += (F[order[2:], k] - F[order[:-2], k]) / span
    return d


class NSGA2:
    def __init__(self, pop_size: int = 72, n_var: int = 1680, crossover_prob: float = 0.9, n_points: int = 4, mutation_prob: float = 0.05,
                 mutation_prob_var: float | None = None, seed: int = 0):
        self.pop_size, self.n_var = pop_size, n_var
        self.crossover_prob, self.n_points, self.mutation_prob = crossover_prob, n_points, mutation_prob
        self.mutation_prob_var = min(0.5, 1.0 / n_var) if mutation_prob_var is None else mutation_prob_var  # pymoo: Mutation.get_prob_var
        self.rng = np.random.default_rng(seed)
        self.n_gen = 0
        self.X = self.F = self.G = None  # survivors
        self._pending = None

    def __setstate__(self, state: dict) -> None:
        """Checkpoints are whole dill pickles of this object (``train_nsga2_base.py``'s ``save_algorithm`` pattern): one written before
        ``mutation_prob_var`` existed must still resume.  It gets pymoo's default and a warning, because the mutation rate it was
        written under (per-bit ``mutation_prob``) differs from the one it continues with."""
        self.__dict__.update(state)
        if "mutation_prob_var" not in state:
            import warnings

            self.mutation_prob_var = min(0.5, 1.0 / self.n_var)
            warnings.warn(f"NSGA2 checkpoint predates mutation_prob_var: resuming with pymoo's default {self.mutation_prob_var:.3g} per bit "
                          f"(it was written under a per-bit rate of {self.mutation_prob})", RuntimeWarning, stacklevel=2)

    def setup(self, X0: np.ndarray | None = None) -> None:
        self._pending = (self.rng.random((self.pop_size, self.n_var)) < 0.5) if X0 is None else np.asarray(X0, dtype=np.bool_)
        self.n_gen = 1

    def ask(self) -> np.ndarray:
        if self._pending is None:
            self._pending = self._offspring()
        return self._pending

    def tell(self, X: np.ndarray, F: np.ndarray, G: np.ndarray) -> None:
        X, F, G = np.asarray(X, dtype=np.bool_), np.asarray(F, dtype=np.float64), np.asarray(G, dtype=np.float64).reshape(-1)
        if self.X is not None:
            X, F, G = np.vstack([self.X, X]), np.vstack([self.F, F]), np.concatenate([self.G, G])
        _, uniq = np.unique(X, axis=0, return_index=True)
        uniq.sort()
        X, F, G = X[uniq], F[uniq], G[uniq]
        keep: list[int] = []
        for front in fast_non_dominated_sort(F, G):
            if len(keep) + len(front) <= self.pop_size:
                keep += front
            else:
                cd = crowding_distance(F[front])
                keep += [front[i] for i in np.argsort(-cd, kind="stable")[: self.pop_size - len(keep)]]
                break
        self.X, self.F, self.G = X[keep], F[keep], G[keep]
        self._rank = np.empty(len(keep), dtype=int)
        for r, front in enumerate(fast_non_dominated_sort(self.F, self.G)):
            self._rank[front] = r
        self._cd = crowding_distance(self.F)
        self._pending = None
        self.n_gen += 1

    def _tournament(self) -> int:
        a, b = self.rng.integers(0, len(self.X), size=2)
        if self._rank[a] != self._rank[b]:
            return a if self._rank[a] < self._rank[b] else b
        return a if self._cd[a] >= self._cd[b] else b

    def _mutate(self, child: np.ndarray) -> np.ndarray:
        """BitflipMutation(prob, prob_var) [ext]: with probability `prob` the individual is mutated, and then each bit flips with `prob_var`."""
        if self.rng.random() <= self.mutation_prob:
            child ^= self.rng.random(self.n_var) < self.mutation_prob_var
        return child

    def _offspring(self) -> np.ndarray:
        seen = {x.tobytes() for x in self.X}
        out: list[np.ndarray] = []
        tries = 0
        while len(out) < self.pop_size and tries < 100 * self.pop_size:
            tries += 1
            p1, p2 = self.X[self._tournament()].copy(), self.X[self._tournament()].copy()
            if self.rng.random() < self.crossover_prob:
                cuts = np.sort(self.rng.choice(np.arange(1, self.n_var), size=self.n_points, replace=False))
                swap = np.zeros(self.n_var, dtype=bool)
                for k in range(0, len(cuts), 2):
                    swap[cuts[k] : (cuts[k + 1] if k + 1 < len(cuts) else self.n_var)] = True
                p1[swap], p2[swap] = p2[swap].copy(), p1[swap].copy()
            for child in (p1, p2):
                child = self._mutate(child)
                key = child.tobytes()
                if key not in seen and len(out) < self.pop_size:  # duplicate elimination, as pymoo's default
                    seen.add(key)
                    out.append(child)
        return np.array(out, dtype=np.bool_)
