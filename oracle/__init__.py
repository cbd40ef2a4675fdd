"""CPU oracle for the DeCafNet grounding hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``cvpr2025-decafnet_amd/`` may import this package.  See DESIGN.md.
"""
