"""Which layers to tap (reference: utils/intermediate_feature_return.py:19-85)."""
from typing import List, Optional, Tuple, Union


class IntermediateFeatureReturner:
    "Mixin holding the intermediate-feature selection."

    def __init__(self, indices: Optional[Union[int, List[int]]] = None, norm_intermediate: bool = True,
                 stop_early: bool = False, intermediates_only: bool = True):
        self.indices = indices
        self.norm_intermediate = norm_intermediate
        self.stop_early = stop_early
        self.intermediates_only = intermediates_only


def feature_take_indices(num_features: int, indices: Optional[Union[int, List[int]]] = None,
                         as_set: bool = False) -> Tuple[List[int], int]:
    """None -> all layers; int n -> the last n; list -> those layers (negative = from the end).
    Returns (absolute indices, max index); raises AssertionError when out of range."""
    if indices is None:
        indices = num_features
    if isinstance(indices, int):
        assert 0 < indices <= num_features, f"last-n ({indices}) is out of range (1 to {num_features})"
        take = list(range(num_features - indices, num_features))
    else:
        take = []
        for i in indices:
            idx = num_features + i if i < 0 else i
            assert 0 <= idx < num_features, f"feature index {idx} is out of range (0 to {num_features - 1})"
            take.append(idx)
    if as_set:
        return set(take), max(take)
    return take, max(take)
