"""BinaryDescriptorMatcher's DATASET form (VERDICT r4 missing #4): add / train / match(query, matches, masks) / knnMatch / radiusMatch over
several train images with imgIdx (ref: /root/reference/src/line_descriptor/src/binary_descriptor_matcher.cpp:70-111, 117-195, 339-425,
508-595).  The device searches the concatenated set; what is checked here is the composition the reference adds around the search --
the image lookup through indexesMap (upper_bound - 1, with std::map::insert's refusal to overwrite the key of an EMPTY image), trainIdx
= the row in the set, the masks applied AFTER the search, compactResult -- against a transcription of those loops over the oracle's
searches (oracle/lf_oracle_lbd.c: the Mihasher tie rule, knn and radius lists)."""
import bisect

import numpy as np
import pytest

from lane_slam_amd import BinaryDescriptorMatcher, FrontEnd, default_config, synth

pytestmark = pytest.mark.gpu


def _flip(code, bits):
    c = code.copy()
    for b in bits:
        c[b >> 3] ^= np.uint8(1 << (b & 7))
    return c


class _RefMatcher(object):
    """The reference's bookkeeping, statement by statement, over the oracle's searches."""

    def __init__(self, o):
        self.o, self.rows, self.index_map, self.next, self.num_images = o, [], {}, 0, 0

    def add(self, descriptors):                                # :70-80
        for d in descriptors:
            self.rows.append(d)
            self.index_map.setdefault(self.next, self.num_images)          # std::map::insert: an existing key is kept
            self.next += d.shape[0]
            self.num_images += 1

    def _img(self, row):                                       # itup = indexesMap.upper_bound(row); itup--
        keys = sorted(self.index_map)
        return self.index_map[keys[bisect.bisect_right(keys, row) - 1]]

    def _set(self):
        return np.concatenate([r.reshape(-1, 32) for r in self.rows])

    def match(self, q, masks):                                 # :117-195
        idx, dist, _ = self.o.match_mih(q, self._set())
        out = []
        for c in range(q.shape[0]):
            if idx[c] < 0:
                continue
            img = self._img(int(idx[c]))
            if not masks or masks[img] is None or masks[img][c] != 0:
                out.append((c, int(idx[c]), img, float(dist[c])))
        return out

    def knn(self, q, k, masks, compact):                       # :339-425
        idx, dist = self.o.knn_match(q, self._set(), k, tie_rule="mihasher")
        lists = []
        for c in range(q.shape[0]):
            tmp = []
            for j in range(k):
                if idx[c, j] < 0:
                    continue
                img = self._img(int(idx[c, j]))
                if not masks or masks[img] is None or masks[img][c] != 0:
                    tmp.append((c, int(idx[c, j]), img, float(dist[c, j])))
            if (len(tmp) == 0 and not compact) or len(tmp) > 0:
                lists.append(tmp)
        return lists

    def radius(self, q, r, masks, compact):                    # :508-595
        off, idx, dist = self.o.radius_match(q, self._set(), r, tie_rule="mihasher")
        lists = []
        for c in range(q.shape[0]):
            tmp = []
            for j in range(off[c], off[c + 1]):
                img = self._img(int(idx[j]))
                if not masks or masks[img] is None or masks[img][c] != 0:
                    tmp.append((c, int(idx[j]), img, float(dist[j])))
            if (len(tmp) == 0 and not compact) or len(tmp) > 0:
                lists.append(tmp)
        return lists


def _t(matches):
    return [(m.queryIdx, m.trainIdx, m.imgIdx, m.distance) for m in matches]


def test_dataset_matcher_against_the_reference_bookkeeping():
    from oracle.oracle import Oracle
    rng = np.random.default_rng(11)
    o = Oracle(default_config("parity"))
    fe = FrontEnd(default_config("parity"), max_frames=1)
    sizes = [300, 0, 157, 1, 0, 0, 420]                         # empty images: the next one is reported under the empty one's number
    images = [synth.random_codes(n, 40 + i) if n else np.zeros((0, 32), np.uint8) for i, n in enumerate(sizes)]
    allrows = np.concatenate(images)
    q = synth.random_codes(200, 99)
    for i in range(60):                                         # near duplicates of set rows (ties, small distances), some exact
        q[i] = _flip(allrows[int(rng.integers(0, allrows.shape[0]))], rng.choice(256, size=int(rng.integers(0, 30)), replace=False))
    images[6][5] = images[0][7]                                  # the same code in two images: a tie at every distance
    q[61] = images[0][7]
    masks = [rng.integers(0, 2, q.shape[0]).astype(np.uint8) for _ in sizes]
    masks[2] = None
    bm, ref = BinaryDescriptorMatcher(fe), _RefMatcher(o)
    bm.add(images[:3]); ref.add(images[:3])
    bm.train()
    bm.add(images[3:]); ref.add(images[3:])                     # a second add after a train: the set is rebuilt from everything
    assert bm.size() == (len(sizes), int(sum(sizes)))
    for mk in (None, masks):
        assert _t(bm.match(q, mk)) == ref.match(q, mk)
        for compact in (False, True):
            got = bm.knnMatch(q, 5, mk, compact)
            want = ref.knn(q, 5, mk, compact)
            assert [_t(l) for l in got] == want
            got = bm.radiusMatch(q, 100.0, mk, compact)
            want = ref.radius(q, 100.0, mk, compact)
            assert [_t(l) for l in got] == want
    used = {m.imgIdx for m in bm.match(q)}
    assert 0 in used and 1 in used and 4 in used and not ({2, 5, 6} & used)       # rows of images 2 and 6 carry their EMPTY predecessors' numbers (1 and 4)
    with pytest.raises(ValueError):
        bm.match(q, masks[:3])
    bm.clear()
    assert bm.size() == (0, 0)
    from lane_slam_amd import LanefrontError
    with pytest.raises(LanefrontError):
        bm.match(q)
    fe.close()
