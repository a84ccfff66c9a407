"""Per-frame segment table: equal-length columns (slot embeddings, ids, scores, class labels ...) selected
together - what the tracker and the post-process of this package pass around.

Role of the reference's `Instances` (mmdet/models/structures/instances.py:11-228, used at
vps_temporal_slots.py:303-346 for `pred_logits` / `pred_masks` / `output_embedding` / `obj_ids`), written for
what this package needs from it: column access by attribute, row selection with a mask / index tensor /
slice / int applied to every column, concatenation of tables, device moves. Columns are torch tensors or
Python lists; a table with no columns has no length.
"""
import torch

_RESERVED = ("_size", "_cols")


def _take(col, sel):
    """Rows `sel` of one column. Lists follow tensor semantics for masks, index tensors and slices."""
    if not isinstance(col, list):
        return col[sel]
    if isinstance(sel, slice):
        return col[sel]
    idx = torch.as_tensor(sel)
    if idx.dtype == torch.bool:
        idx = idx.nonzero().flatten()
    return [col[int(i)] for i in idx.reshape(-1)]


class Instances:
    def __init__(self, image_size, **columns):
        self.__dict__["_size"] = tuple(image_size)
        self.__dict__["_cols"] = {}
        for name, col in columns.items():
            self.set(name, col)

    # ---- columns ------------------------------------------------------------------------------------
    @property
    def image_size(self):
        return self._size

    def set(self, name, col):
        rows = len(col)
        if self._cols and rows != len(self):
            raise AssertionError(f"column '{name}' has {rows} rows, the table has {len(self)}")
        self._cols[name] = col

    def has(self, name):
        return name in self._cols

    def remove(self, name):
        self._cols.pop(name)

    def get_fields(self):
        return self._cols

    def __setattr__(self, name, col):
        if name in _RESERVED:
            self.__dict__[name] = col
        else:
            self.set(name, col)

    def __getattr__(self, name):          # only reached when normal lookup fails, i.e. for column names
        cols = self.__dict__.get("_cols", {})
        if name in cols:
            return cols[name]
        raise AttributeError(f"no column '{name}' in this segment table (columns: {sorted(cols)})")

    def __len__(self):
        if not self._cols:
            raise NotImplementedError("a segment table without columns has no length")
        return len(next(iter(self._cols.values())))

    def __repr__(self):
        n = len(self) if self._cols else 0
        return f"Instances(rows={n}, image_size={self._size}, columns={sorted(self._cols)})"

    # ---- row selection / movement -----------------------------------------------------------------
    def _like(self, columns):
        out = Instances(self._size)
        for name, col in columns:
            out.set(name, col)
        return out

    def __getitem__(self, sel):
        if isinstance(sel, int):                       # a single row stays a table of one row
            n = len(self)
            if not -n <= sel < n:
                raise IndexError(f"row {sel} of a segment table with {n} rows")
            sel = sel % n
            sel = slice(sel, sel + 1)
        return self._like((name, _take(col, sel)) for name, col in self._cols.items())

    def to(self, *args, **kwargs):
        return self._like((name, col.to(*args, **kwargs) if isinstance(col, torch.Tensor) else col)
                          for name, col in self._cols.items())

    @staticmethod
    def cat(tables):
        tables = list(tables)
        if not tables:
            raise AssertionError("nothing to concatenate")
        first = tables[0]
        if len(tables) == 1:
            return first
        out = Instances(first.image_size)
        for name, col in first.get_fields().items():
            parts = [t.get_fields()[name] for t in tables]
            if isinstance(col, torch.Tensor):
                out.set(name, torch.cat(parts, dim=0))
            elif isinstance(col, list):
                out.set(name, [x for part in parts for x in part])
            else:
                raise ValueError(f"column '{name}': cannot concatenate {type(col).__name__}")
        return out
