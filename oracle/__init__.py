"""CPU oracle for the GConv-Adapter hot path — TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code. Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the timed CPU baseline. The product
package (``gconv-adapter_amd/``) never imports this package and has no CPU
compute path.

PARITY UNPINNED: the arithmetic of the reference's hot path lives in the
un-vendored third-party dependency ``torch-geometric`` (pinned ``==2.5.3`` in
``requirements/mpnn_inductive_environment.yml:102`` and ``==1.7.2`` in
``requirements/gtn_transductive_environment.yaml:91`` of the reference), which
is not installed in the build container and cannot be installed (no network).
The reference repository holds no tests, golden vectors or fixtures for this
path. This oracle therefore restates the published ``GCNConv`` / ``gcn_norm`` /
``MessagePassing.propagate`` semantics and anchors on the reference's own call
sites (``src/finetune/gconv_adapter.py:40-41,73-78,92-106``). Three independent
derivations are cross-checked against each other and against the committed golden vectors:
``pyg_restated.py`` (edge lists, PyG's op order, torch autograd), ``dense_ref.py`` (dense fp64
matrix form, hand-written backward) and ``gconv_oracle.c`` (plain C, scalar loops over the edge
list, hand-written backward; compiled by ``__graft_entry__.build()``, loaded through
``c_oracle.py``).
"""
