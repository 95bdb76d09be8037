"""CPU oracle for the sliding-window inference path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (deep_contact_estimator_amd) never does.
"""
