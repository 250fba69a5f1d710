"""od_wscl_amd -- MI355X-native (gfx950) implementation of OD-WSCL's
proposal-feature hot path, behind the reference's `wetectron.layers` /
`wetectron.modeling` operator API.  See DESIGN.md."""
__version__ = "0.1.0"

