#!/usr/bin/env python3
"""Bytes a rank receives per propagation product under the two operand exchanges of the row-partitioned graph trainers
(qrec_amd/graph.py::_setup_row_exchange): all-gather of every block vs only the remote rows the rank's block of the adjacency
refers to (dist.RowPartition.reference).  Host arithmetic on the two synthetic Yelp2018-shape graphs, world 8, d = 64 -- no
GPU needed; prints one JSON object (committed as profiles/r03_graph_exchange_bytes.json; round 4 adds the batch-row economies: profiles/r04_graph_exchange_bytes.json)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hostkern as HK
from qrec_amd import dist as qd
from qrec_amd.graph import joint_norm_adjacency, spectral_row_key
from qrec_amd.synth import make_dataset


class FakeComm:
    def __init__(self, world, rank): self.world, self.rank = world, rank


def bytes_for(adj, n, ld, world):
    per_rank = []
    for r in range(world):
        rp = qd.RowPartition(FakeComm(world, r), n, ld, kern=HK)
        rp.reference(adj[0], adj[1])
        per_rank.append(dict(rows_referenced=int(rp.ref_recv_rows.sum()), bytes_received=int(rp.ref_recv_rows.sum() - rp.ref_recv_rows[r]) * ld * 4))
    return per_rank


out = {"world": 8, "ld": 64, "graphs": {}}
for shape in ("yelp2018", "yelp2018-clustered"):
    d = make_dataset(shape); nu, ni = d["n_users"], d["n_items"]; n = nu + ni
    adj = joint_norm_adjacency(nu, ni, d["train_u"], d["train_i"])
    allgather = (8 - 1) * (-(-n // 8)) * 64 * 4
    res = {"allgather_bytes_received_per_product": allgather, "as_numbered": bytes_for(adj, n, 64, 8)}
    # the same graph with users and items renumbered along the spectral key (communities become id ranges); the partition is
    # still [users ; items] cut into 8 contiguous blocks, so a block of user rows still needs item rows from the item ranks
    key = spectral_row_key(adj[0], adj[1], adj[2], nu)
    pu, pi = np.argsort(key[:nu], kind="stable"), np.argsort(key[nu:], kind="stable")
    inv_u, inv_i = np.empty(nu, np.int64), np.empty(ni, np.int64); inv_u[pu] = np.arange(nu); inv_i[pi] = np.arange(ni)
    adj2 = joint_norm_adjacency(nu, ni, inv_u[d["train_u"]], inv_i[d["train_i"]])
    res["spectral_renumbered"] = bytes_for(adj2, n, 64, 8)
    for k in ("as_numbered", "spectral_renumbered"):
        b = [x["bytes_received"] for x in res[k]]
        res[k + "_summary"] = {"max_MB": max(b) / 1e6, "mean_MB": float(np.mean(b)) / 1e6, "vs_allgather": float(np.mean(b)) / allgather}
    L = 3
    res["lightgcn_L3_step_MB_per_rank"] = {"allgather_form": (2 * L + 1) * (allgather / 1e6),
                                           "referenced_as_numbered": 2 * L * res["as_numbered_summary"]["mean_MB"] + allgather / 1e6,
                                           "referenced_spectral": 2 * L * res["spectral_renumbered_summary"]["mean_MB"] + allgather / 1e6,
                                           "note": "2L propagation products + one all-gather of the layer sum for the batch loss (every rank evaluates the whole batch)"}
    # round 4 -- the batch-row economies of the row-partitioned steps (graph.RowPartitionedLightGCNTrainer / NGCFTrainer, batch_rows=True):
    # the layer sum travels as ONE all-reduce of the batch's 3B rows (ring: 2 (G-1)/G x payload received per rank), the first backward
    # product takes the batch gradient every rank already holds (no exchange); NGCF: the two all-gathers of the normalised blocks
    # become one all-reduce of 3B rows of the 3d-wide table (padded to 256 floats)
    B, G = 2048, 8
    small = 2 * (G - 1) / G * 3 * B * 64 * 4 / 1e6
    wide = 2 * (G - 1) / G * 3 * B * 256 * 4 / 1e6
    ag, ref = allgather / 1e6, res["as_numbered_summary"]["mean_MB"]
    res["round4_batch_rows"] = {
        "batch": B, "allreduce_3B_rows_MB_received": small, "allreduce_3B_wide_rows_MB_received": wide,
        "lightgcn_L3_step_MB_per_rank": {"allgather_form": {"round3": (2 * L + 1) * ag, "round4": (2 * L - 1) * ag + small},
                                         "referenced_as_numbered": {"round3": 2 * L * ref + ag, "round4": (2 * L - 1) * ref + small}},
        "ngcf_step_MB_per_rank": {"allgather_form": {"round3": 6 * ag, "round4": 4 * ag + wide},
                                  "referenced_as_numbered": {"round3": 3 * ag + 3 * ref, "round4": ag + 3 * ref + wide},
                                  "note": "round 3: E_0 gathered whole (operand of layer 1 and ego block), z_1 and z_2 all-gathered, 3 more operand exchanges (E_1 forward, "
                                          "dside of both layers backward); round 4: the z gathers replaced by the 3B-row all-reduce"},
        "simgcl_L2_step_MB_per_rank": {"allgather_form": {"round3": 9 * ag, "round4": 5 * ag + 2 * (G - 1) / G * (3 * B + 2 * 3900) * 64 * 4 / 1e6},
                                       "referenced_as_numbered": {"round3": 6 * ref + 3 * ag, "round4": 5 * ref + 2 * (G - 1) / G * (3 * B + 2 * 3900) * 64 * 4 / 1e6},
                                       "note": "round 3: 4 forward operand exchanges (the shared first product + 3), three all-gathers of the layer sums Sm / S1 / S2, 2 backward; "
                                               "round 4: the three gathers become ONE all-reduce of [Sm: 3B | S1, S2: the batch's ~3,900 unique users + items] rows and the first "
                                               "backward product reads the batch gradient every rank holds (no exchange)"}}
    out["graphs"][shape] = res
print(json.dumps(out, indent=1))
