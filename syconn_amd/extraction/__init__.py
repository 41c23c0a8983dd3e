"""Device-side successors of ``syconn.extraction`` stages that consume the dense predictions."""
