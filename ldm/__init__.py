"""`ldm` — import-path compatibility layer over upgpt_amd.

The reference wires everything through `instantiate_from_config({target: "ldm.…", params})`
(ldm/util.py:78-93), so the dotted paths in configs/deepfashion/bbox.yaml are part of the
drop-in contract.  These modules only re-export the MI355X-native
implementations; there is no code from the reference here.
"""
