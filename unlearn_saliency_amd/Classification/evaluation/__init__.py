"""SVC membership-inference attack used for the "MIA" column (reference
Classification/evaluation/SVC_MIA.py:85-150).  CPU sklearn evaluation — not part of the
accelerated path — kept so `main_random.py` reports the same metrics."""
from .svc_mia import SVC_MIA
