"""Minimal stand-ins for ``torch_geometric.data.Data`` / ``Batch``.

The reference only ever reads ``x / edge_index / edge_attr / batch`` from these objects and calls
``Batch.from_data_list`` / ``to_data_list`` (modeling_llamole.py:328-333, 611-616, 829-834), so the hot path
does not need PyG at run time.  A real PyG ``Data``/``Batch`` is accepted anywhere these are (duck typing).
"""
from __future__ import annotations

from typing import List, Optional

import torch


class GraphData:
    def __init__(self, x, edge_index, edge_attr, num_nodes: Optional[int] = None):
        self.x, self.edge_index, self.edge_attr = x, edge_index, edge_attr
        self.num_nodes = int(num_nodes if num_nodes is not None else x.size(0))

    def to(self, device):
        self.x, self.edge_index, self.edge_attr = self.x.to(device), self.edge_index.to(device), self.edge_attr.to(device)
        return self


class GraphBatch:
    def __init__(self, x, edge_index, edge_attr, batch, sizes: List[int]):
        self.x, self.edge_index, self.edge_attr, self.batch = x, edge_index, edge_attr, batch
        self._sizes = list(sizes)
        self.num_graphs = len(self._sizes)
        self.batch._ll_num_graphs = self.num_graphs      # lets the GIN wrappers skip reading batch[-1] back from the device

    @classmethod
    def from_data_list(cls, data_list):
        xs, eis, eas, bs, sizes, off = [], [], [], [], [], 0
        for g, d in enumerate(data_list):
            n = int(d.x.size(0))
            xs.append(d.x)
            eis.append(d.edge_index + off)
            eas.append(d.edge_attr)
            bs.append(torch.full((n,), g, dtype=torch.long, device=d.x.device))
            sizes.append(n)
            off += n
        return cls(torch.cat(xs), torch.cat(eis, dim=1), torch.cat(eas), torch.cat(bs), sizes)

    def to_data_list(self):
        out, off = [], 0
        for n in self._sizes:
            sel = (self.edge_index[0] >= off) & (self.edge_index[0] < off + n)
            out.append(GraphData(self.x[off:off + n], self.edge_index[:, sel] - off, self.edge_attr[sel], n))
            off += n
        return out

    def __len__(self):
        return self.num_graphs

    def __getitem__(self, idx):
        """PyG's `Batch[idx]`: an int gives one graph, an index tensor / list / slice the list of selected graphs (the reference
        indexes the retro product batch with the valid-label indices, modeling_llamole.py:396-399)."""
        graphs = self.to_data_list()
        if isinstance(idx, int):
            return graphs[idx]
        if isinstance(idx, slice):
            return graphs[idx]
        if torch.is_tensor(idx):
            idx = idx.nonzero().view(-1).tolist() if idx.dtype == torch.bool else idx.view(-1).tolist()
        return [graphs[int(i)] for i in idx]

    def to(self, device):
        self.x, self.edge_index = self.x.to(device), self.edge_index.to(device)
        self.edge_attr, self.batch = self.edge_attr.to(device), self.batch.to(device)
        self.batch._ll_num_graphs = self.num_graphs
        return self
