#!/usr/bin/env python3
"""Extract the reference's known-answer data into small JSON fixtures.

Run in the build container only (it reads /root/reference, which does not exist on the GPU
box); the JSON it writes is committed.  Sources (all *recorded outputs / data files*, no
reference source text is copied):

  * examples/intro.ipynb  cells 3, 7, 10, 11, 12, 16, 24, 29, 35 (recorded outputs)
  * examples/models/{perm_square_3x3,lf_5_line,clifford_3q_custom}.json (env configs)
  * examples/models/{perm_square_3x3,lf_5_line,clifford_3q_custom}.pt   (the trained policies the reference ships:
    plain state dicts of tensors, read with torch.load(weights_only=True) and stored as float16 .npz under policies/)

The reference has no tests; these notebook transcripts are the only
executable-derived vectors that exist for this path.
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def cell_outputs(nb, idx):
    cell = nb["cells"][idx]
    texts = []
    for o in cell.get("outputs", []):
        if "text" in o:
            texts.append("".join(o["text"]))
        elif "data" in o and "text/plain" in o["data"]:
            texts.append("".join(o["data"]["text/plain"]))
    return "".join(cell["source"]), "\n".join(texts)


def parse_matrices(text):
    """All '[[a b c]\n [d e f]...]' blocks (numpy print) or 'array([[a, b],..' blocks."""
    mats = []
    for m in re.finditer(r"\[\[[0-9,\s\[\]]+?\]\]", text):
        rows = re.findall(r"\[([0-9,\s]+)\]", m.group(0).replace("[[", "[").replace("]]", "]"))
        mats.append([[int(x) for x in re.findall(r"\d+", r)] for r in rows])
    return mats


def parse_gate_list(text):
    return [[name, [int(a), int(b)]] for name, a, b in re.findall(r"\('(\w+)', \((\d+), (\d+)\)\)", text)]


def convert_policies():
    """The reference's trained BasicPolicy checkpoints (embeddings / common.0 / action.0 / value.0, weight + bias) as
    float16 arrays.  They are the strongest known-answer data the reference holds for this path: a policy trained
    against the reference's env solves random targets only on an env with the same observation layout, action order,
    gate semantics and solved test (tests/test_reference_policies.py)."""
    import numpy as np
    import torch

    os.makedirs(os.path.join(OUT, "policies"), exist_ok=True)
    for name in ("perm_square_3x3", "lf_5_line", "clifford_3q_custom"):
        sd = torch.load(os.path.join(REF, "examples/models", name + ".pt"), map_location="cpu", weights_only=True)
        arrays = {k.replace(".", "_"): v.numpy().astype(np.float16) for k, v in sd.items()}
        np.savez_compressed(os.path.join(OUT, "policies", name + ".npz"), **arrays)


def main():
    convert_policies()
    nb = json.load(open(os.path.join(REF, "examples/intro.ipynb")))

    # ---- gateset orderings -------------------------------------------------------------
    gatesets = {}
    _, out3 = cell_outputs(nb, 3)
    gatesets["lf_line3_bidirectional_cx_swap"] = {
        "source": "examples/intro.ipynb cell 3 output (LinearFunctionGym.from_coupling_map(CouplingMap.from_line(3, bidirectional=True)))",
        "env": "linear_function",
        "edges": [[0, 1], [1, 0], [1, 2], [2, 1]],
        "basis_gates": None,
        "gateset": parse_gate_list(out3),
    }
    _, out16 = cell_outputs(nb, 16)
    gatesets["perm_grid3x3_unidirectional_swap"] = {
        "source": "examples/intro.ipynb cell 16 output (PermutationGym.from_coupling_map(CouplingMap.from_grid(3,3, bidirectional=False)))",
        "env": "permutation",
        # CouplingMap.from_grid(3, 3, bidirectional=False): right and down neighbours, row-major ids
        "edges": [[0, 3], [0, 1], [1, 4], [1, 2], [2, 5], [3, 6], [3, 4], [4, 7], [4, 5], [5, 8], [6, 7], [7, 8]],
        "basis_gates": None,
        "gateset": parse_gate_list(out16),
    }
    for name in ("perm_square_3x3", "lf_5_line", "clifford_3q_custom"):
        cfg = json.load(open(os.path.join(REF, "examples/models", name + ".json")))
        gatesets["model_" + name] = {
            "source": f"examples/models/{name}.json",
            "env_cls": cfg["env_cls"],
            "env": cfg["env"],
        }
    json.dump(gatesets, open(os.path.join(OUT, "gatesets.json"), "w"), indent=1)

    # ---- LinearFunction line-3 state-transition transcripts -----------------------------
    gs = gatesets["lf_line3_bidirectional_cx_swap"]["gateset"]
    _, out7 = cell_outputs(nb, 7)
    start = parse_matrices(out7)[0]
    seqs = []
    src10, out10 = cell_outputs(nb, 10)
    m10 = parse_matrices(out10)[0]
    fin10 = out10.strip().endswith("True)")
    seqs.append({"source": "intro.ipynb cell 10", "actions": [2], "states": [m10], "is_final": [fin10]})
    for idx in (11, 12):
        src, out = cell_outputs(nb, idx)
        actions = [int(a) for a in re.search(r"for a in \[([0-9,]+)\]", src).group(1).split(",")]
        mats = parse_matrices(out)
        finals = [s == "True" for s in re.findall(r"Is final: (True|False)", out)]
        assert mats[0] == start and len(mats) == len(actions) + 1 and len(finals) == len(actions)
        seqs.append({"source": f"intro.ipynb cell {idx}", "actions": actions, "states": mats[1:], "is_final": finals})
    lf = {
        "note": "Rewards printed in the notebook come from an older reward scheme and contradict the current "
                "metrics.rs (SURVEY.md section 4); they are deliberately NOT part of this fixture.",
        "num_qubits": 3,
        "gateset": gs,
        "start_state": start,
        "start_source": "intro.ipynb cell 7 (env.set_state(env.get_state(qc)), qc = cx(0,2))",
        "sequences": seqs,
        "obs_dtype": "int8",
        "action_space_n": 8,
        "obs_shape": [3, 3],
    }
    json.dump(lf, open(os.path.join(OUT, "lf_line3_transcripts.json"), "w"), indent=1)

    # ---- recorded synthesis outputs: input state + the circuit the reference printed -----
    # The gate sequences are hand-transcribed from the recorded circuit drawings (kept verbatim
    # below as `drawing`); gates drawn in one column commute, so their relative order is free.
    _, draw24 = cell_outputs(nb, 24)
    _, draw29 = cell_outputs(nb, 29)
    _, draw35 = cell_outputs(nb, 35)
    sols = {
        "permutation_swap_0_8": {
            "source": "intro.ipynb cells 23-24: rls.synth(QuantumCircuit(9).swap(0,8)) on the 3x3 grid",
            "env": "permutation",
            "num_qubits": 9,
            "gateset": gatesets["perm_grid3x3_unidirectional_swap"]["gateset"],
            # get_state = argsort(permutation_pattern) (envs/synthesis.py:295-303); swap(0,8) is an involution
            "state": [8, 1, 2, 3, 4, 5, 6, 7, 0],
            "circuit": [["SWAP", [0, 1]], ["SWAP", [7, 8]], ["SWAP", [4, 7]], ["SWAP", [1, 4]],
                        ["SWAP", [0, 1]], ["SWAP", [4, 7]], ["SWAP", [7, 8]]],
            "drawing": draw24,
        },
        "linear_function_cx_0_4": {
            "source": "intro.ipynb cells 28-30: rls.synth(QuantumCircuit(5).cx(0,4)) on the 5-line, CX only; "
                      "cell 30 records LinearFunction(in) == LinearFunction(out) -> True",
            "env": "linear_function",
            "num_qubits": 5,
            "gateset": gatesets["model_lf_5_line"]["env"]["gateset"],
            # get_state = LinearFunction(Clifford(qc).adjoint()).linear (envs/synthesis.py:254-258); same shape as
            # the recorded 3-qubit case of cell 7: identity plus entry (target, control)
            "state": [[1, 0, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, 1, 0], [1, 0, 0, 0, 1]],
            "circuit": [["CX", [0, 1]], ["CX", [4, 3]], ["CX", [1, 2]], ["CX", [3, 4]], ["CX", [2, 3]],
                        ["CX", [1, 2]], ["CX", [0, 1]], ["CX", [1, 2]], ["CX", [2, 3]], ["CX", [1, 2]],
                        ["CX", [3, 4]], ["CX", [4, 3]]],
            "drawing": draw29,
        },
        "clifford_h_2": {
            "source": "intro.ipynb cell 35: rls.synth(QuantumCircuit(3).h(2)) with the custom 3q gateset",
            "env": "clifford",
            "num_qubits": 3,
            "gateset": gatesets["model_clifford_3q_custom"]["env"]["gateset"],
            # get_state = Clifford(qc).adjoint().tableau[:, :-1].T.flatten() (envs/synthesis.py:206-209): for H on
            # qubit 2 the phase-less tableau is the identity with rows 2 and 5 exchanged (symmetric, self-adjoint)
            "state": [[1, 0, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1],
                      [0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 1, 0], [0, 0, 1, 0, 0, 0]],
            "circuit": [["SWAP", [1, 2]], ["SWAP", [0, 1]], ["H", [0]], ["SWAP", [0, 1]], ["SWAP", [1, 2]]],
            "drawing": draw35,
        },
    }
    json.dump(sols, open(os.path.join(OUT, "notebook_solutions.json"), "w"), indent=1, ensure_ascii=False)
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".json")))


if __name__ == "__main__":
    main()
