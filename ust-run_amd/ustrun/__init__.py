"""ustrun -- host side of the MI355X U-Net training hot path (ctypes over libustrun.so)."""
