import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import ttn_amd as T
from oracle import tt_oracle as O
from helpers import to_oracle, to_product, tt_rel_diff
T.ensure_init(0)
rng = np.random.default_rng(3)
for d, rks in ((1, [1, 1]), (2, [1, 2, 1]), (2, [1, 1, 1]), (3, [1, 2, 2, 1]), (3, [1, 1, 1, 1])):
    x = O.rand_tt((2,) * d, rks, rng)
    for c in range(1, d + 1):
        got = T.orthogonalize(to_product(x), i=c); ref = O.orthogonalize(x, i=c)
        print(d, rks, c, list(got.ttv_rks) == ref.ttv_rks, list(got.ttv_ot) == ref.ttv_ot, np.linalg.norm(O.ttv_to_tensor(to_oracle(got)) - O.ttv_to_tensor(x)))
