

def test_size_requests_above_the_worst_case_are_clamped():
    """a stock MuJoCo file's <size nconmax="100" njmax="500"/> must load -- a request above what the model can ever
    produce is clamped to the worst case (with a compile note), a request below it is honoured."""
    from mujoco_ros_pkgs_amd import mjcf
    xml = """
<mujoco><size nconmax="100" njmax="500"/>
<option cone="pyramidal" solver="PGS"/>
<worldbody><geom name="floor" type="plane" size="1 1 0.1"/>
  <body pos="0 0 0.1"><freejoint/><geom type="box" size="0.05 0.05 0.05" mass="1"/></body></worldbody></mujoco>"""
    m = mjcf.compile_xml_string(xml)
    assert m["nconmax"] == 4 and m["nefcmax"] == 16           # plane - box: 4 contacts x 4 pyramidal rows
    assert len(m["compile_notes"]) == 2 and "njmax=500" in m["compile_notes"][0]
    m2 = mjcf.compile_xml_string(xml.replace('nconmax="100" njmax="500"', 'nconmax="2" njmax="6"'))
    assert (m2["nconmax"], m2["nefcmax"]) == (2, 6) and m2["compile_notes"] == []
    from mujoco_ros_pkgs_amd import engine
    engine.CompiledModel(m)  # mjb_compile accepts it (it refused nefcmax = 500 under PGS)
