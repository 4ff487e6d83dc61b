"""CPU oracle of the value-iteration hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may
import this package; the product (stodynprog_amd/) never does.
"""
